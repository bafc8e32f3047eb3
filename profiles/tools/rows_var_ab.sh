#!/bin/bash
export DGTTA_LIB="$(dirname "$(readlink -f "$0")")/libdgtta_hip_diag.so"      # laboratory build (python -m dg_tta_amd.build --diag): the product library has no *_ABL / ROWS_VAR switches
# same-box A/B of the row-reuse kernel's feature masks (DGTTA_ROWS_VAR): single-layer timings + cycle stamps
out=gpurun_out/rows_var_ab.txt
: > $out
for v in 0 1 3 5 7 0 7; do
  for c in "32 32 128" "64 32 128" "64 64 64"; do
    echo -n "VAR=$v " >> $out
    DGTTA_ROWS_VAR=$v python profiles/tools/kbench.py conv bf16 $c 40 2>/dev/null >> $out
  done
done
for v in 0 3 7; do
  echo "---- stamps VAR=$v" >> $out
  DGTTA_ROWS_VAR=$v python profiles/tools/rows_stamps.py 32 32 128 2>/dev/null >> $out
done
cat $out
