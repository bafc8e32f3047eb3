#!/bin/bash
# per-kernel times of the feature kernels (rocprofv3 kernel trace): usage prof_feat.sh <size> <batch> <what>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_feat
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_feat -o pf --output-format csv -- python3 $R/scratch/featbench.py $1 $2 $3 > $R/gpurun_out/prof_feat.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_feat/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
