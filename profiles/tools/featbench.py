"""Micro-bench of the feature / loss kernels at 128^3 (per-sample microseconds): GIN chain, MIND, warps, loss."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import ops
from dg_tta_amd.gin import draw_gin_params
from dg_tta_amd.mind import MIND3D
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
what = sys.argv[3].split(",") if len(sys.argv) > 3 else ["gin", "mind", "mind_bf16"]

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in evs)[iters // 2] * 1e3 / B

x = torch.randn(B, 1, n, n, n, device=DEV)
if "gin" in what:
    for ks_force in (None, [3, 3, 3, 3], [1, 1, 1, 1]):
        torch.manual_seed(0)
        alpha, ks, kers, shifts = draw_gin_params(B, DEV)
        if ks_force is not None:
            chans = [1, 2, 2, 2, 1]
            ks = ks_force
            kers = [torch.randn(chans[i + 1] * B, chans[i], k, k, k, device=DEV) for i, k in enumerate(ks)]
        t = timeit(lambda: ops.gin_chain(x, alpha, ks, kers, shifts))
        print(f"gin_chain ks={ks} {n}^3 x{B}: {t:8.1f} us per sample ({16 * n**3 / t / 1e6:.2f} TB/s algorithmic)")
if "mind" in what:
    noise = torch.randn(B, 12, n, n, n, device=DEV)
    m = MIND3D()
    t = timeit(lambda: m(x, noise))
    print(f"mind3d fp32 out (noise input) {n}^3 x{B}: {t:8.1f} us per sample")
if "mind_bf16" in what:
    noise = torch.randn(B, 12, n, n, n, device=DEV)
    m = MIND3D()
    t = timeit(lambda: m(x, noise, out_dtype=torch.bfloat16))
    print(f"mind3d bf16 out (noise input) {n}^3 x{B}: {t:8.1f} us per sample")
    t = timeit(lambda: torch.randn(B, 12, n, n, n, device=DEV))
    print(f"torch.randn noise draw       {n}^3 x{B}: {t:8.1f} us per sample")
