"""Micro-bench of single kernels through the C ABI (for rocprofv3 --pmc runs). usage: kbench.py [conv|dgrad|wgrad] [bf16|fp16|fp32] cin cout n iters [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
what, dts, cin, cout, n, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
B = int(sys.argv[7]) if len(sys.argv) > 7 else 1
dt = {"bf16": 1, "fp16": 2, "fp32": 0}[dts]
tdt = {1: torch.bfloat16, 2: torch.float16, 0: torch.float32}[dt]
DEV = "cuda:0"
x = torch.randn(B, n, n, n, cin, device=DEV).to(tdt)
if what == "conv":
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
    S = int(os.environ.get("KB_STRIDE", "1"))          # KB_STRIDE=2: the encoder transitions (n = input edge)
    no = (n - 1) // S + 1
    y = torch.empty((B, no, no, no, cout), dtype=tdt, device=DEV)
    st = None
    if os.environ.get("KB_STATS"):
        st = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, cout, no, no, no), dtype=torch.uint8, device=DEV)
    run = lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, ptr(st) if st is not None else None, B, cin, cout, cin, cout, n, n, n, S, dt, 2, stream_of()), "fwd")
elif what == "dgrad":
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
    dy = torch.randn(B, n, n, n, cout, device=DEV).to(tdt)
    dx = torch.empty((B, n, n, n, cin), dtype=tdt, device=DEV)
    run = lambda: check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wpack), ptr(dx), cin, B, cin, cout, cin, cout, n, n, n, 1, 0, dt, 2, stream_of()), "dgrad")
else:
    dy = torch.randn(B, n, n, n, cout, device=DEV).to(tdt)
    dw = torch.empty((cout, cin, 3, 3, 3), device=DEV)
    # KB_SPLIT=1 (fp32): the split workspace -> six 16-bit launches on three-term bf16 splits
    nb = (lib.dgtta_conv3d_wgrad_split_ws_bytes if os.environ.get("KB_SPLIT") else lib.dgtta_conv3d_wgrad_ws_bytes)(B, cin, cout, n, n, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    run = lambda: check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cin, ptr(dy), cout, ptr(dw), None, ptr(ws), nb, B, cin, cout, n, n, n, 1, 0, dt, 2, stream_of()), "wgrad")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
vox = ((n - 1) // int(os.environ.get("KB_STRIDE", "1")) + 1) ** 3 if what == "conv" else n ** 3
print(f"{what} {dts} {cin}->{cout} {n}^3 x{B}: {ms:.3f} ms, {B*2*27*cin*cout*vox/ms/1e9:.1f} TFLOP/s")
