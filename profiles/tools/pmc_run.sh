#!/bin/bash
# PMC passes (separate runs per counter, kernel-trace only) for the two dominant kernels at the 128^3 32->32 layer
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for what in conv wgrad; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    KB_STATS=1 rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/pmc_${what}_${ctr} -o pmc --output-format csv -- python3 $R/profiles/tools/kbench.py $what bf16 32 32 128 12 > $R/gpurun_out/pmc_${what}_${ctr}.log 2>&1 || exit 1
  done
done
cd $R
python profiles/tools/pmc_summary.py > gpurun_out/pmc_summary.json
