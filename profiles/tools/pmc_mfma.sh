#!/bin/bash
# MFMA-utilisation / LDS / wait counters (rocprofv3 --pmc, one group per run, --kernel-trace only) of the MFMA kernels on the
# TIMED launch shape of bench.py: fp16 storage, 8 samples per launch (2 branches x 4 accumulation steps), 128^3.
# usage (inside gpurun): bash profiles/tools/pmc_mfma.sh <tag> ; writes gpurun_out/<tag>_mfma_util.json
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/${tag}_counters_list.txt 2>&1 || true
groups=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
        "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
        "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "GRBM_GUI_ACTIVE"
        "FETCH_SIZE"
        "WRITE_SIZE")
IFS=";" read -ra JOBS <<< "${PMC_JOBS:-conv fp16 32 32 128 6 8;dgrad fp16 32 32 128 6 8;conv fp16 64 32 128 6 8;wgrad fp16 32 32 128 6 8;conv fp16 128 128 32 10 8;conv fp32 32 32 128 3 8}"
for job in "${JOBS[@]}"; do
  set -- $job
  name=$1_$3_$4_$5
  [ "$2" != fp16 ] && name=$1_$2_$3_$4_$5
  i=0
  for grp in "${groups[@]}"; do
    i=$((i+1))
    # PMC_GROUPS="1 7 8": only these counter groups (1 = MFMA busy, 7 = FETCH_SIZE, 8 = WRITE_SIZE)
    if [ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $i "* ]]; then continue; fi
    KB_STATS=1 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/${tag}_pmc_${name}_$i -o pmc --output-format csv -- python3 $R/profiles/tools/kbench.py $job > $R/gpurun_out/${tag}_pmc_${name}_$i.log 2>&1 || echo "$name group $i ($grp) failed"
  done
  echo "done $name"
done
cd $R
python3 profiles/tools/pmc_mfma_summary.py $tag > gpurun_out/${tag}_mfma_util.json
