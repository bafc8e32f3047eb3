#!/bin/bash
# usage (inside gpurun): bash profiles/tools/fetch_calib.sh ; prints FETCH_SIZE per launch of the three access patterns against the 2 GiB each reads
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/profiles/tools/fetch_calib.hip || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fc -o fc --output-format csv -- /tmp/fetch_calib > /tmp/fc.log 2>&1 || { tail -5 /tmp/fc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
t, n = collections.Counter(), collections.Counter()
for f in glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0]
            t[k] += float(r["Counter_Value"]); n[k] += 1
for k in t:
    per = t[k] / n[k] * 1024
    print(f"{k:28s} FETCH_SIZE per launch {per/2**30:6.3f} GiB raw = {per/2**31:5.3f} of the 2 GiB read ({n[k]} launches)")
PY
