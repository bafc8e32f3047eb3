#!/bin/bash
# usage (inside gpurun): bash profiles/tools/prof_epoch.sh <tag> [bench args]; writes gpurun_out/<tag>_stats.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o $tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32 --inference-size 0 "$@" > $R/gpurun_out/$tag.log 2>&1 || exit 1
cd $R
python - "$tag" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows)
out = [f"total kernel s {tot/1e9:.4f}  per epoch ms {tot/4e6:.2f} (4 epochs incl. warmup)"]
for r in rows[:45]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    out.append(f"{n:70s} calls {r['Calls']:>6s} tot/ep {int(r['TotalDurationNs'])/4e6:7.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
open(f'gpurun_out/{tag}_stats.txt', 'w').write('\n'.join(out) + '\n')
PY
