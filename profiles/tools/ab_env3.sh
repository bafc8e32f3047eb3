#!/bin/bash
# same-box A/B/C of one environment switch over whole epochs: ab_env3.sh VAR "v1 v2 v3" [bench args]; prints epochs/s, ms per epoch
var=$1; vals=$2; shift 2
for rep in 1 2; do
for v in $vals; do
  env $var=$v python bench.py --no-fp32 --inference-size 0 --no-cpu-baseline --steps ${AB_STEPS:-4} --warmup 2 "$@" > gpurun_out/ab_env.log 2>&1
  python - "$var" "$v" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_env.log") if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_ms"], flush=True)
PY
done
done
