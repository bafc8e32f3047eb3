#!/bin/bash
# conv kernel ablations at 128^3 32->32 bf16 (diagnostic)
for a in 0 1 2 3 4 5 6 7; do
  if [ $a = 0 ]; then unset DGTTA_CONV_ABL; else export DGTTA_CONV_ABL=$a; fi
  echo -n "ABL=$a: "; python scratch/kbench.py conv bf16 32 32 128 50 | tail -1
done
