#!/bin/bash
for v in 1 2 3 0; do echo "DGTTA_CONVT_GEMM=$v"; DGTTA_CONVT_GEMM=$v python scratch/compbench.py 8 2>&1 | grep "^convT" | head -2; done
