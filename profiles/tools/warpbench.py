"""Micro-bench of the logit warps at the bench's launch shape (8 x 128^3 x 16 fp32, channels-last): forward + backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import ops
from dg_tta_amd.tta.augmentation_utils import get_rand_affine
DEV = "cuda:0"
B, C, N = 8, 16, 128
torch.manual_seed(0)
x = torch.randn(B, N, N, N, C, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
_, rinv = get_rand_affine(B)
rinv = rinv.float().contiguous().to(DEV)
gy = torch.randn(B, N, N, N, C, device=DEV).permute(0, 4, 1, 2, 3)
def run():
    y = ops.affine_warp(x, rinv, padding_mode="zeros", tta_grid_algebra=True)
    y.backward(gy)
    x.grad = None
for _ in range(2): run()
torch.cuda.synchronize()
import time
fw = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = {"fwd": 0.0, "bwd": 0.0}
for _ in range(10):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); y = ops.affine_warp(x, rinv, padding_mode="zeros", tta_grid_algebra=True); e[1].record()
    y.backward(gy); e[2].record(); torch.cuda.synchronize(); x.grad = None
    t["fwd"] += e[0].elapsed_time(e[1]) / 10; t["bwd"] += e[1].elapsed_time(e[2]) / 10
gb = 2 * B * N ** 3 * C * 4 / 1e9
print(f"fwd {t['fwd']:.3f} ms ({gb / t['fwd']:.2f} TB/s), bwd {t['bwd']:.3f} ms ({gb / t['bwd']:.2f} TB/s)")
