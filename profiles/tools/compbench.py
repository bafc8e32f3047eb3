import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load(); DEV = "cuda:0"; dt = 1; tdt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = {}
for (cin, cout, n) in ((64, 32, 64), (128, 64, 32), (256, 128, 16), (320, 256, 8)):
    # convT cin->cout from n^3 to (2n)^3
    x = torch.randn(B, n, n, n, cin, device=DEV).to(tdt); w = torch.randn(cin, cout, 2, 2, 2, device=DEV) * 0.1; b = torch.zeros(cout, device=DEV)
    out = torch.empty((B, 2*n, 2*n, 2*n, cout), dtype=tdt, device=DEV)
    nb = lib.dgtta_convT3d_fwd_ws_bytes(cin, cout, dt); ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    t_f = timeit(lambda: check(lib.dgtta_convT3d_k2s2_fwd(ptr(x), cin, ptr(w), ptr(b), ptr(out), cout, ptr(ws), nb, B, cin, cout, n, n, n, dt, 0, stream_of()), "f"))
    dout = torch.randn_like(out); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    nb2 = lib.dgtta_convT3d_bwd_ws_bytes(B, cin, cout, n, n, n); ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
    t_b = timeit(lambda: check(lib.dgtta_convT3d_k2s2_bwd(ptr(x), cin, ptr(dout), cout, ptr(w), ptr(dx), cin, ptr(dw), ptr(db), ptr(ws2), nb2, B, cin, cout, n, n, n, 0, dt, 0, stream_of()), "b"))
    print(f"convT {cin}->{cout} {n}^3: fwd {t_f*1e3:.0f} us  bwd {t_b*1e3:.0f} us")
    tot["convT_f"] = tot.get("convT_f", 0) + t_f; tot["convT_b"] = tot.get("convT_b", 0) + t_b
for (cin, cout, n) in ((32, 64, 128), (64, 128, 64), (128, 256, 32), (256, 320, 16)):
    x = torch.randn(B, n, n, n, cin, device=DEV).to(tdt); w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wp = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // 2, dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wp), cin, cout, cin, cout, dt, stream_of()), "p")
    m = n // 2
    y = torch.empty((B, m, m, m, cout), dtype=tdt, device=DEV); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    t_f = timeit(lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wp), None, ptr(y), cout, None, B, cin, cout, cin, cout, n, n, n, 2, dt, 0, stream_of()), "f"))
    t_d = timeit(lambda: check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wp), ptr(dx), cin, B, cin, cout, cin, cout, n, n, n, 2, 0, dt, 0, stream_of()), "d"))
    nb = lib.dgtta_conv3d_wgrad_ws_bytes(B, cin, cout, m, m, m); ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    t_w = timeit(lambda: check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cin, ptr(dy), cout, ptr(dw), None, ptr(ws), nb, B, cin, cout, n, n, n, 2, 0, dt, 0, stream_of()), "w"))
    print(f"s2conv {cin}->{cout} {n}^3: fwd {t_f*1e3:.0f} us  dgrad {t_d*1e3:.0f} us  wgrad {t_w*1e3:.0f} us")
    for k, v in (("s2_f", t_f), ("s2_d", t_d), ("s2_w", t_w)): tot[k] = tot.get(k, 0) + v
print({k: round(v, 3) for k, v in tot.items()}, "ms per branch pass; per epoch: fwd x33, bwd x32")
print("epoch ms:", 33 * (tot["convT_f"] + tot["s2_f"]) + 32 * (tot["convT_b"] + tot["s2_d"] + tot["s2_w"]))
