#!/bin/bash
# round-end evidence: default bench line, the same command under rocprofv3 --kernel-trace --stats, fp32 line
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/final_bench_default.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof -o r01 --output-format csv -- python $R/bench.py > $R/gpurun_out/final_bench_under_profiler.log 2>&1 || exit 1
cd $R && python bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/final_bench_fp32.log 2>&1 || exit 1
tail -1 gpurun_out/final_bench_default.log | cut -c1-400
