#!/bin/bash
# kernel-level times behind a layerbench.py table: rocprofv3 --kernel-trace --stats over the same launches
# usage (on the GPU box): LB_LAYERS=enc4.1,dec0.0 bash profiles/tools/layer_kernels.sh <tag> [fp16|bf16] [batch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
tag=${1:-lk}; dt=${2:-fp16}; B=${3:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_lk -o lk --output-format csv -- python3 $R/profiles/tools/layerbench.py $dt $B > $R/gpurun_out/${tag}_lk.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_lk.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/${tag}_lk/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(f"{int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:130]}")
PY
