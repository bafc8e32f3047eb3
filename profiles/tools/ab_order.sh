#!/bin/bash
# same-box A/B of the row-reuse conv's job order (DGTTA_ROWS_ORDER 1 = round robin in compact blocks, 0 = contiguous ranges)
for o in 1 0 1 0; do
  DGTTA_ROWS_ORDER=$o python bench.py --no-fp32 --inference-size 0 --no-cpu-baseline --steps 3 > gpurun_out/ab_order_$o.log 2>&1
  python - <<PY
import json; d=json.loads(open("gpurun_out/ab_order_$o.log").read().strip().splitlines()[-1]); print("order $o", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_ms"])
PY
done
