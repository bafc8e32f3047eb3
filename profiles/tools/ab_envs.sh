#!/bin/bash
# same-box A/B of two ENVIRONMENTS over whole epochs, alternating: ab_envs.sh "VAR1=a VAR2=b" "VAR1=c" [repeats]
# (an empty string = the default environment); prints ms per epoch of every run
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for E in "$A" "$B"; do
    env $E python bench.py --no-fp32 --inference-size 0 --no-cpu-baseline --steps ${AB_STEPS:-4} ${AB_ARGS:-} > gpurun_out/ab_envs.log 2>&1
    python - "$E" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_envs.log") if l.startswith("{")][-1])
print(f"[{sys.argv[1] or 'default'}] {d['ms_per_step']} ms per epoch, {d['value']} epochs/s ({d['dtype']})", flush=True)
PY
  done
done
