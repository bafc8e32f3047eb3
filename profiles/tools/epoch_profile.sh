#!/bin/bash
# Whole-epoch evidence for bench.py's `epoch_roofline`: HBM bytes per TTA epoch and time per kernel name.
# Per storage type, three pairs of runs of the SAME bench command, with 1 and with 3 epochs (no warm-up, He weights: traffic and
# time do not depend on the weights), so that (run3 - run1) / 2 is exactly one steady-state epoch without the set-up:
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace      (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace
#   rocprofv3 --kernel-trace --stats               (ONE stream: DGTTA_WGRAD_STREAM=0 DGTTA_PIPELINE_PREP=0 - with the default side
#                                                     streams overlapped kernels wait for CUs inside their own duration and the
#                                                     per-name times no longer add up to the epoch: 283 ms instead of 170)
# usage (inside gpurun): bash profiles/tools/epoch_profile.sh <tag> ["bf16 fp32"] ["fetch write stats"] ; writes gpurun_out/<tag>_epoch_profile.json
tag=${1:-r05}
dts=${2:-fp16}
parts=${3:-"fetch write stats"}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dt in $dts; do
  for n in 1 3; do
    common="--dtype $dt --steps $n --warmup 0 --weights he --no-fp32 --no-cpu-baseline --inference-size 0"
    [[ " $parts " != *" fetch "* ]] || rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_ep_${dt}_fetch_$n -o ep --output-format csv -- python3 $R/bench.py $common > $R/gpurun_out/${tag}_ep_${dt}_fetch_$n.log 2>&1 || echo "fetch $dt $n failed"
    [[ " $parts " != *" write "* ]] || rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_ep_${dt}_write_$n -o ep --output-format csv -- python3 $R/bench.py $common > $R/gpurun_out/${tag}_ep_${dt}_write_$n.log 2>&1 || echo "write $dt $n failed"
    [[ " $parts " != *" stats "* ]] || DGTTA_WGRAD_STREAM=0 DGTTA_PIPELINE_PREP=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_ep_${dt}_stats_$n -o ep --output-format csv -- python3 $R/bench.py $common > $R/gpurun_out/${tag}_ep_${dt}_stats_$n.log 2>&1 || echo "stats $dt $n failed"
    echo "done $dt $n"
  done
done
cd $R
python3 profiles/tools/epoch_profile_summary.py $tag $dts > gpurun_out/${tag}_epoch_profile.json
python3 profiles/tools/dec31_dispatches.py $tag $(echo $dts | cut -d" " -f1) > gpurun_out/${tag}_dec31_dispatches.csv 2> gpurun_out/${tag}_dec31_dispatches.txt || true
# the raw csv files are large: keep the per-name statistics only
for dt in $dts; do
  for n in 1 3; do
    f=$(find gpurun_out/${tag}_ep_${dt}_stats_$n -name '*kernel_stats.csv' 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f gpurun_out/${tag}_epoch_${dt}_${n}ep_kernel_stats.csv
    rm -rf gpurun_out/${tag}_ep_${dt}_fetch_$n gpurun_out/${tag}_ep_${dt}_write_$n gpurun_out/${tag}_ep_${dt}_stats_$n
  done
done
