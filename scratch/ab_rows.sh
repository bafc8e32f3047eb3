#!/bin/bash
# timing of the row-reuse kernel on three layer shapes, 3 rounds (run-to-run spread inside one box)
for round in 1 2 3; do
  for shape in "32 32 128" "64 64 64" "64 32 128"; do
    echo -n "round $round: "; KB_STATS=1 python scratch/kbench.py conv bf16 $shape 30 2>&1 | tail -1
  done
done
