#!/bin/bash
# SQ counters of the feature kernels: usage pmc_feat.sh <size> <batch> <what> "<counters>"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_feat
rocprofv3 --pmc $4 --kernel-trace -d $R/gpurun_out/pmc_feat -o pf --output-format csv -- python3 $R/scratch/featbench.py $1 $2 $3 > $R/gpurun_out/pmc_feat.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_feat/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "mind_ssd" in k or "gin_chain" in k or "mind_finish" in k:
        print(k)
        for c, v in d.items():
            v = sorted(v)
            print(f"   {c:28s} median {v[len(v)//2]:.4g}  (n={len(v)})")
PY
