#!/bin/bash
for a in 0 1 2 3 4 5; do
  if [ $a = 0 ]; then unset DGTTA_ROWS_ABL; else export DGTTA_ROWS_ABL=$a; fi
  echo -n "ROWS_ABL=$a: "; python scratch/kbench.py conv bf16 32 32 128 50 | tail -1
done
