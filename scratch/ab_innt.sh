#!/bin/bash
for o in 1 0 1 0; do
  DGTTA_IN_NT=$o python bench.py --no-fp32 --inference-size 0 --no-cpu-baseline --steps 3 > gpurun_out/ab_innt_$o.log 2>&1
  python - <<PY
import json; d=json.loads(open("gpurun_out/ab_innt_$o.log").read().strip().splitlines()[-1]); print("in_nt $o", d["value"], d["ms_per_step"])
PY
done
