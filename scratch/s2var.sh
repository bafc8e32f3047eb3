#!/bin/bash
# stride-2 forward tile variants (DGTTA_CONV_S2) on the two large layers
for v in 0 2 3 5; do
  for cfg in "32 64 128" "64 128 64"; do
    echo -n "S2=$v  "; DGTTA_CONV_S2=$v python scratch/s2bench.py $cfg 20 2>&1 | grep "s2 conv"
  done
done
