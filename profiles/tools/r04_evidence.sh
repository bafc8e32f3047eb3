#!/bin/bash
# Round-4 evidence in two gpurun calls (inside gpurun, from the repo root): bash profiles/tools/r04_evidence.sh 1|2
# part 1: PMC passes on the timed launch shape, rocprofv3 --stats of the bf16 / fp32 epochs and the inference leg, cycle stamps
# part 2: same-box A/B of the round's kernels, the head+accumulate ablations, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
part=${1:-1}
if [ "$part" = 1 ]; then
  bash profiles/tools/pmc_mfma.sh r04 > $O/r04_pmc.log 2>&1 || exit 1
  echo "pmc done"
  DGTTA_WGRAD_STREAM=0 DGTTA_PIPELINE_PREP=0 bash profiles/tools/prof_epoch.sh r04bf16 || exit 1
  echo "bf16 profile done"
  DGTTA_WGRAD_STREAM=0 DGTTA_PIPELINE_PREP=0 bash profiles/tools/prof_epoch.sh r04fp32 --dtype fp32 || exit 1
  echo "fp32 profile done"
  bash profiles/tools/prof_infer.sh r04inf 512 bf16 || exit 1
  echo "inference profile done"
  python3 profiles/tools/ring_stamps.py > $O/r04_ring_stamps.txt 2>&1 || exit 1
  python3 profiles/tools/wring_clock.py > $O/r04_wring_clock.txt 2>&1 || exit 1
  echo "stamps done"
else
  ab() {   # ab <label> [ENV=VALUE ...]: 8 timed epochs of the bf16 product path with the given switches
    label=$1; shift
    line=$(env "$@" python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --no-parity --inference-size 0 2>/dev/null | grep '^{' | tail -1)
    echo "$label $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], "ms/epoch", d["value"], "epochs/s; dec.3.1", r["kernel"], r["avg_ms"], "ms", r["frac"])')"
  }
  {
    echo "# same box, one after the other; bf16 storage, 8 timed epochs each"
    ab "default                " DGTTA_NOP=1
    ab "DGTTA_CONV_RING=0      " DGTTA_CONV_RING=0
    ab "DGTTA_WGRAD_RING=0     " DGTTA_WGRAD_RING=0
    ab "both off (round 3 path)" DGTTA_CONV_RING=0 DGTTA_WGRAD_RING=0
    ab "DGTTA_CONV_RING=1 (all)" DGTTA_CONV_RING=1
    ab "default again          " DGTTA_NOP=1
    echo "# head + Gaussian accumulate of one 128^3 x 105 window (profiles/tools/habench.py)"
    for t in fp32 fp16; do
      echo "MFMA   $(python3 profiles/tools/habench.py $t)"
      echo "FMA    $(DGTTA_HA_MFMA=0 python3 profiles/tools/habench.py $t)"
      echo "no logits (accumulator traffic only) $(DGTTA_HA_ABL=1 python3 profiles/tools/habench.py $t)"
      echo "no accumulator traffic (logits only) $(DGTTA_HA_ABL=2 python3 profiles/tools/habench.py $t)"
    done
    echo "# fused head + inverse logit warp, 8 x 128^3 (profiles/tools/headwarpbench.py); second block: every gather read L1 resident (DGTTA_WARP_ABL=1)"
    HW_DT=bf16 python3 profiles/tools/headwarpbench.py
    DGTTA_LIB=profiles/tools/libdgtta_hip_diag.so DGTTA_WARP_ABL=1 HW_DT=bf16 python3 profiles/tools/headwarpbench.py
    echo "# InstanceNorm backward of one layer (profiles/tools/inbench.py)"
    python3 profiles/tools/inbench.py 32 128 8
    python3 profiles/tools/inbench.py 64 64 8
  } > $O/r04_ab.txt 2>&1
  echo "ab done"
  python3 bench.py > $O/r04_bench_default.log 2> $O/r04_bench_default.err || exit 1
  echo "bench done"
fi
