#!/bin/bash
# usage (inside gpurun): bash profiles/tools/prof_infer.sh <tag> [volume] [dtype]; rocprofv3 --stats of the sliding-window inference leg alone
# (bench.inference_leg: one ensemble member, 128^3 windows at step 0.5); writes gpurun_out/<tag>_stats.txt
tag=$1; vol=${2:-512}; dt=${3:-bf16}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o $tag --output-format csv -- python3 $R/profiles/tools/infer_only.py $vol $dt > $R/gpurun_out/$tag.log 2>&1 || exit 1
cd $R
python - "$tag" <<'PY'
import csv, sys, json
tag = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/{tag}/{tag}_kernel_stats.csv')))
line = [l for l in open(f'gpurun_out/{tag}.log') if l.startswith('{')][-1]
nwin = json.loads(line)["windows"]
tot = sum(int(r['TotalDurationNs']) for r in rows)
out = [line.strip(), f"total kernel s {tot/1e9:.4f}  per window ms {tot/1e6/(nwin+1):.3f} ({nwin} windows + 1 warm-up window)"]
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    out.append(f"{n:70s} calls {r['Calls']:>6s} tot {int(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%")
open(f'gpurun_out/{tag}_stats.txt', 'w').write('\n'.join(out) + '\n')
PY
