#!/bin/bash
# A/B of the row-reuse kernel inside ONE box: default build vs DGTTA_ROWS_ABL=<n> variants, interleaved, 3 rounds
for round in 1 2 3; do
  for a in 0 $@; do
    if [ $a = 0 ]; then unset DGTTA_ROWS_ABL; else export DGTTA_ROWS_ABL=$a; fi
    echo -n "round $round ABL=$a: "; KB_STATS=1 python scratch/kbench.py conv bf16 32 32 128 50 2>&1 | tail -1
  done
done
