"""InstanceNorm backward of one layer through the C ABI (reduce + finalize + apply): python3 profiles/tools/inbench.py [C] [size] [batch]
(run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from dg_tta_amd import _lib, ops
from dg_tta_amd._lib import check, ptr, stream_of
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
lib = _lib.load()
V = n ** 3
y = torch.randn((B, V, C), device=dev).to(torch.bfloat16)
gz = torch.randn((B, V, C), device=dev).to(torch.bfloat16)
dy = torch.empty_like(gz)
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
mr = torch.stack([torch.randn(B, C, device=dev) * 0.1, torch.rand(B, C, device=dev) + 0.5], -1).contiguous()
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
nb = lib.dgtta_instnorm_ws_bytes(B, C, V)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    check(lib.dgtta_instnorm_lrelu_bwd(ptr(gz), C, ptr(y), C, ptr(gamma), ptr(beta), ptr(mr), ptr(dy), C, ptr(dg), ptr(db), ptr(ws), nb,
                                       B, C, V, 0.01, 1, ops.BF16, stream_of(dev)), "in_bwd")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
gb = B * V * C * 2 * 5 / 1e9
print(f"instnorm bwd {B}x{n}^3x{C}: {ms*1e3:.1f} us, {gb/ms:.2f} TB/s over 5 tensor passes ({gb:.2f} GB)")

# forward: statistics + finalize + apply (y -> z) of the same layer
z = torch.empty_like(y)
mr2 = torch.empty((B, C, 2), device=dev)
def runf():
    check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), C, None, ptr(gamma), ptr(beta), ptr(mr2), ptr(z), C, ptr(ws), nb, B, C, V, 1e-5, 0.01, ops.BF16,
                                       stream_of(dev)), "in_fwd")
for _ in range(3): runf()
torch.cuda.synchronize()
e0.record()
for _ in range(20): runf()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
gb = B * V * C * 2 * 3 / 1e9
print(f"instnorm fwd {B}x{n}^3x{C}: {ms*1e3:.1f} us, {gb/ms:.2f} TB/s over 3 tensor passes ({gb:.2f} GB)")
