"""Micro-bench of the stride-2 conv forward through the C ABI: s2bench.py cin cout n iters"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
cin, cout, n, iters = [int(a) for a in sys.argv[1:5]]
DEV = "cuda:0"
x = torch.randn(1, n, n, n, cin, device=DEV).bfloat16()
w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, 1) // 2, dtype=torch.bfloat16, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, 1, stream_of()), "pack")
m = n // 2
y = torch.empty((1, m, m, m, cout), dtype=torch.bfloat16, device=DEV)
run = lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, None, 1, cin, cout, cin, cout, n, n, n, 2, 1, 2, stream_of()), "fwd")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"s2 conv bf16 {cin}->{cout} {n}^3->{m}^3: {ms:.3f} ms, {2*27*cin*cout*m**3/ms/1e9:.1f} TFLOP/s")
