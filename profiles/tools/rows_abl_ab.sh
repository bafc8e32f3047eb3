#!/bin/bash
export DGTTA_LIB="$(dirname "$(readlink -f "$0")")/libdgtta_hip_diag.so"      # laboratory build (python -m dg_tta_amd.build --diag): the product library has no *_ABL / ROWS_VAR switches
# timing models of the row-reuse kernel (results are WRONG in these builds): 1 no DMA, 3 no MFMA, 4 no A DMA, 5 no weight DMA,
# 9 only 4 of the 6 D-planes of the A tile are fetched (what a ring along D would fetch)
out=gpurun_out/rows_abl_ab.txt
: > $out
for a in 0 1 3 4 5 9 0; do
  for c in "32 32 128" "64 32 128" "64 64 64"; do
    echo -n "ABL=$a " >> $out
    DGTTA_ROWS_VAR=0 DGTTA_ROWS_ABL=$a python profiles/tools/kbench.py conv bf16 $c 40 2>/dev/null >> $out
  done
done
cat $out
