#!/bin/bash
# Round-6 evidence on the final tree (each call < 20 min; every step writes under gpurun_out/):
#   bash profiles/tools/r06_evidence.sh A   two default bench lines + the fp32 trajectory an N > 1 line compares with
#   bash profiles/tools/r06_evidence.sh B   whole-epoch traffic (PMC) + per-name time (one stream) + dec.3.1 dispatch rows, fp16 and fp32
#   bash profiles/tools/r06_evidence.sh C   PMC of the probed kernels on the timed shape (MFMA busy, FETCH, WRITE) + inference profile
#   bash profiles/tools/r06_evidence.sh D   BASELINE config 2's 12 x 16 schedule refereed by the CPU oracle (64^3), every storage type
R=$GRAFT_REPO_ROOT
cd $R
case "$1" in
A)
  for i in 1 2; do
    ( time python3 bench.py ) > gpurun_out/r06_bench_line_$i.log 2>&1
    grep '^{' gpurun_out/r06_bench_line_$i.log | tail -1 > gpurun_out/r06_bench_line_$i.json
    python3 - <<PY
import json
d = json.load(open("gpurun_out/r06_bench_line_$i.json"))
r = d["roofline"]
print("line $i:", d["value"], "epochs/s", d["ms_per_step"], "ms", d["dtype"], "; roofline", r["kernel"], r["frac"], "(in schedule", r["in_schedule"]["frac"], "); forward", r["forward"]["frac"],
      "; epoch", r["epoch_frac"], "; fp32", d["fp32"]["value"], "; bf16", d["bf16"]["value"],
      "; dice_delta", {k: (d["dice_delta"][k]["loss"], round(d["dice_delta"][k]["hard_dice"], 6), d["dice_delta"][k]["within_tolerance"]) for k in ("fp32", "fp16", "bf16")},
      "; at size", {k: d["dice_delta"]["at_headline_size"][k]["hard_dice_vs_gt_after"] for k in ("fp32", "fp16", "bf16")}, "from", d["dice_delta"]["at_headline_size"]["fp32"]["hard_dice_vs_gt_before"],
      "; inference", d["inference"]["seconds"], "s (logits fp32", d["inference"]["fp32_logits_accumulator"]["seconds"], "s); cpu", d["cpu_baseline"]["value"])
PY
  done
  python3 bench.py --write-fp32-trajectory 8 > gpurun_out/r06_traj.log 2>&1; tail -1 gpurun_out/r06_traj.log
  cp profiles/fp32_trajectory.json gpurun_out/r06_fp32_trajectory.json
  ;;
B)
  bash profiles/tools/epoch_profile.sh r06 "fp16 fp32" "fetch write stats" > gpurun_out/r06_epoch_profile.log 2>&1
  tail -4 gpurun_out/r06_epoch_profile.log; cat gpurun_out/r06_dec31_dispatches.txt
  ;;
C)
  PMC_GROUPS="1 7 8" PMC_JOBS="conv fp16 32 32 128 6 8;wgrad fp16 32 32 128 6 8;conv fp32 32 32 128 3 8" bash profiles/tools/pmc_mfma.sh r06 > gpurun_out/r06_pmc.log 2>&1
  tail -3 gpurun_out/r06_pmc.log; head -c 600 gpurun_out/r06_mfma_util.json
  rm -rf gpurun_out/r06_pmc_*_[0-9]
  bash profiles/tools/prof_infer.sh r06inf 512 fp16; head -14 gpurun_out/r06inf_stats.txt; rm -rf gpurun_out/r06inf
  python3 profiles/tools/layerbench.py fp16 8 > gpurun_out/r06_layerbench.txt 2>&1; tail -3 gpurun_out/r06_layerbench.txt
  ;;
D)
  lr=${2:-3e-4}      # (3e-4: the referee's rate; 1e-5: the plan's own, config_log_utils.py:26)
  python3 profiles/tools/referee_12_epochs.py $lr > gpurun_out/r06_referee12.log 2>&1
  python3 - <<PY
import json
t = open("gpurun_out/r06_referee12.log").read()
d = json.loads(t[t.rindex("\n{"):])
json.dump(d, open("gpurun_out/r06_dice_delta_12_epochs.json", "w"), indent=1)
print("fp32_drift", d.get("fp32_drift"))
for k in ("fp32", "fp16", "bf16"):
    e = d[k]
    print(k, "loss", e["loss"], "tol", e["loss_tolerance"], "pseudo", e["pseudo_dice"], "hard", e["hard_dice"], "labels", e["label_agreement"], "within", e["within_tolerance"])
PY
  ;;
esac
