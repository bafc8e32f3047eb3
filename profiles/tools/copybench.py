"""What a plain device copy reaches on this box, beside the InstanceNorm forward apply pass on the same bytes
(8 x 128^3 x 32 x 2 B = 1.07 GB read + 1.07 GB written): python3 profiles/tools/copybench.py"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from dg_tta_amd import _lib, ops
from dg_tta_amd._lib import check, ptr, stream_of
dev = torch.device("cuda:0")
lib = _lib.load()
B, V, C = 8, 128 ** 3, 32
xs = [torch.randn((B, V, C), device=dev).to(torch.bfloat16) for _ in range(3)]
ys = [torch.empty_like(x) for x in xs]
def timeit(fn, n=30):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = 2 * B * V * C * 2 / 1e9
t = timeit(lambda i: ys[i % 3].copy_(xs[i % 3]))
print(f"torch copy_            {t*1e3:7.1f} us  {gb/t:.2f} TB/s")
xf = [x.view(torch.float32) for x in xs]; yf = [y.view(torch.float32) for y in ys]
t = timeit(lambda i: torch.add(xf[i % 3], 1.0, out=yf[i % 3]))
print(f"torch add (fp32 view)  {t*1e3:7.1f} us  {gb/t:.2f} TB/s")
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
mr = torch.zeros((B, C, 2), device=dev)
nb = lib.dgtta_instnorm_ws_bytes(B, C, V)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
stats = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, C, 128, 128, 128), dtype=torch.uint8, device=dev)
def fwd(i):
    check(lib.dgtta_instnorm_lrelu_fwd(ptr(xs[i % 3]), C, None, ptr(gamma), ptr(beta), ptr(mr), ptr(ys[i % 3]), C, ptr(ws), nb, B, C, V, 1e-5, 0.01,
                                       ops.BF16, stream_of(dev)), "in_fwd")
t = timeit(fwd)
print(f"instnorm fwd (stats pass + apply: 3 tensor passes) {t*1e3:7.1f} us  {1.5*gb/t:.2f} TB/s")
