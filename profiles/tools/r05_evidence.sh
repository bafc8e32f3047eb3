#!/bin/bash
# Round-5 evidence on the final tree, in three gpurun calls (each < 20 min; every step appends to gpurun_out/):
#   bash profiles/tools/r05_evidence.sh A   two default bench lines + the fp32 trajectory an N > 1 line compares with
#   bash profiles/tools/r05_evidence.sh B   whole-epoch traffic (PMC) + per-name time (one stream) + dec.3.1 dispatch rows, bf16 and fp32
#   bash profiles/tools/r05_evidence.sh C   PMC of the dominant kernels on the timed shape (MFMA busy, FETCH, WRITE) + inference profile
R=$GRAFT_REPO_ROOT
cd $R
case "$1" in
A)
  for i in 1 2; do
    ( time python3 bench.py ) > gpurun_out/r05_bench_line_$i.log 2>&1
    grep '^{' gpurun_out/r05_bench_line_$i.log | tail -1 > gpurun_out/r05_bench_line_$i.json
    python3 - <<PY
import json
d = json.load(open("gpurun_out/r05_bench_line_$i.json"))
print("line $i:", d["value"], "epochs/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], "; epoch", d["epoch_roofline"]["frac"], "; fp32", d["fp32"]["value"],
      "; dice_delta", {k: (round(d["dice_delta"][k]["hard_dice"], 6), d["dice_delta"][k]["label_agreement"]) for k in ("fp32", "fp16", "bf16")},
      "; inference", d["inference"]["seconds"], "s (logits fp32", d["inference"]["fp32_logits_accumulator"]["seconds"], "s); cpu", d["cpu_baseline"]["value"])
PY
  done
  python3 bench.py --write-fp32-trajectory 8 > gpurun_out/r05_traj.log 2>&1; tail -1 gpurun_out/r05_traj.log
  cp profiles/fp32_trajectory.json gpurun_out/r05_fp32_trajectory.json
  ;;
B)
  bash profiles/tools/epoch_profile.sh r05 "bf16 fp32" "fetch write stats" > gpurun_out/r05_epoch_profile.log 2>&1
  tail -4 gpurun_out/r05_epoch_profile.log; cat gpurun_out/r05_dec31_dispatches.txt
  python3 - <<PY
import json
d = json.load(open("gpurun_out/r05_epoch_profile.json"))
for k, e in d.items():
    print(k, e["kernel_ms_per_epoch"], "ms;", round(e["hbm_bytes_per_epoch"] / 1e9, 1), "GB;", e["largest_consumer"])
PY
  ;;
C)
  PMC_GROUPS="1 7 8" PMC_JOBS="conv fp16 32 32 128 6 8;wgrad fp16 32 32 128 6 8;conv fp32 32 32 128 3 8" bash profiles/tools/pmc_mfma.sh r05 > gpurun_out/r05_pmc.log 2>&1
  tail -3 gpurun_out/r05_pmc.log; head -c 600 gpurun_out/r05_mfma_util.json
  rm -rf gpurun_out/r05_pmc_*_[0-9]
  bash profiles/tools/prof_infer.sh r05inf 512 bf16; head -14 gpurun_out/r05inf_stats.txt; rm -rf gpurun_out/r05inf
  ;;
esac
