#!/bin/bash
# same-box A/B of one environment switch over whole epochs: ab_env.sh VAR val_a val_b   (prints epochs/s, ms per epoch)
var=$1; a=$2; b=$3
for v in $a $b $a $b; do
  env $var=$v python bench.py --no-fp32 --inference-size 0 --no-cpu-baseline --steps ${AB_STEPS:-3} > gpurun_out/ab_env.log 2>&1
  python - "$var" "$v" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_env.log") if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["achieved"])
PY
done
