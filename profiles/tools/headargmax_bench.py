"""Head + argmax from feature-space accumulators, timed on a quarter of a 512^3 volume and scaled: python3 profiles/tools/headargmax_bench.py
(DGTTA_FEATURE_HEAD_MFMA=0: the vector-ALU kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dg_tta_amd import ops
V = 512 * 512 * 512 // 4
for M in (1, 3):
    facc = torch.randn(M, 1, 1, V, 32, device="cuda")
    nsum = torch.rand(1, 1, V, device="cuda") + 0.5
    w = torch.randn(M, 105, 32, device="cuda")
    b = torch.randn(105, device="cuda")
    ops.feature_head_argmax(facc, nsum, w, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.feature_head_argmax(facc, nsum, w, b)
    torch.cuda.synchronize()
    print(f"DGTTA_FEATURE_HEAD_MFMA={os.environ.get('DGTTA_FEATURE_HEAD_MFMA', '1')} members {M}: {(time.perf_counter() - t0) / 3 * 4 * 1e3:.2f} ms per 512^3 volume")
