"""Micro-bench of single kernels through the C ABI (for rocprofv3 --pmc runs). usage: kbench.py [conv|wgrad] [bf16|fp32] cin cout n iters"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
what, dts, cin, cout, n, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
dt = 1 if dts == "bf16" else 0
tdt = torch.bfloat16 if dt else torch.float32
DEV = "cuda:0"
x = torch.randn(1, n, n, n, cin, device=DEV).to(tdt)
if what == "conv":
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
    y = torch.empty((1, n, n, n, cout), dtype=tdt, device=DEV)
    import os
    st = None
    if os.environ.get("KB_STATS"):
        st = torch.zeros(lib.dgtta_conv3d_stats_bytes(1, cout, n, n, n), dtype=torch.uint8, device=DEV)
    run = lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, ptr(st) if st is not None else None, 1, cin, cout, cin, cout, n, n, n, 1, dt, 2, stream_of()), "fwd")
else:
    dy = torch.randn(1, n, n, n, cout, device=DEV).to(tdt)
    dw = torch.empty((cout, cin, 3, 3, 3), device=DEV)
    nb = lib.dgtta_conv3d_wgrad_ws_bytes(1, cin, cout, n, n, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    run = lambda: check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cin, ptr(dy), cout, ptr(dw), None, ptr(ws), nb, 1, cin, cout, n, n, n, 1, 0, dt, 2, stream_of()), "wgrad")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{what} {dts} {cin}->{cout} {n}^3: {ms:.3f} ms, {2*27*cin*cout*n**3/ms/1e9:.1f} TFLOP/s")
