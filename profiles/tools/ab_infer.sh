#!/bin/bash
# same-box A/B of two environments on the sliding-window inference leg (512^3, one member): ab_infer.sh "ENV_A" "ENV_B" [repeats]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for E in "$A" "$B"; do
    env $E python bench.py --no-fp32 --no-cpu-baseline --steps 1 --warmup 0 --referee-epochs 0 --weights he > gpurun_out/ab_infer.log 2>&1
    python - "$E" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_infer.log") if l.startswith("{")][-1])
i = d["inference"]
print(f"[{sys.argv[1] or 'default'}] feature-space {i['seconds']} s ({i['ms_per_window']} ms/window, {i['frac_of_mfma_peak']} of peak); fp32 logits accumulator {i['fp32_logits_accumulator']['seconds']} s", flush=True)
PY
  done
done
