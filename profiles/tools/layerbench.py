"""Per-layer table of the 3d_fullres net at the bench's launch shape (8 samples per launch, 16-bit storage): forward (with fused
statistics), data gradient and weight gradient of every 3x3x3 conv through the C ABI, ~0.15 s of back-to-back launches each
(sustained clocks), TFLOP/s against the 2.5 PF peak and the time a layer loses against the ring kernels' 0.5 of peak.
usage: layerbench.py [fp16|bf16] [batch]      (LB_LAYERS=enc4.1,dec0.0,...: only these rows)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
dts = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = {"bf16": 1, "fp16": 2}[dts]
tdt = {1: torch.bfloat16, 2: torch.float16}[dt]
DEV = "cuda:0"
# (name, cin, cout, input edge, stride) of the 18 conv blocks at a 128^3 patch (plans.json:279-401; unet.PLANS_3D_FULLRES: five
# stages, features 32 64 128 256 320)
LAYERS = [("enc0.0", 12, 32, 128, 1), ("enc0.1", 32, 32, 128, 1), ("enc1.0", 32, 64, 128, 2), ("enc1.1", 64, 64, 64, 1),
          ("enc2.0", 64, 128, 64, 2), ("enc2.1", 128, 128, 32, 1), ("enc3.0", 128, 256, 32, 2), ("enc3.1", 256, 256, 16, 1),
          ("enc4.0", 256, 320, 16, 2), ("enc4.1", 320, 320, 8, 1),
          ("dec0.0", 512, 256, 16, 1), ("dec0.1", 256, 256, 16, 1), ("dec1.0", 256, 128, 32, 1), ("dec1.1", 128, 128, 32, 1),
          ("dec2.0", 128, 64, 64, 1), ("dec2.1", 64, 64, 64, 1), ("dec3.0", 64, 32, 128, 1), ("dec3.1", 32, 32, 128, 1)]
# LB_EXTRA=1: shapes of a six-stage plan as well (NOT in this network; until round 6 this table listed them as if they were)
if os.environ.get("LB_EXTRA"):
    LAYERS += [("x.8^3-s2", 320, 320, 8, 2), ("x.4^3", 320, 320, 4, 1), ("x.cat8^3", 640, 320, 8, 1)]


def timed(run, budget_ms=150.0):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    n = max(3, min(400, int(budget_ms / max(e0.elapsed_time(e1), 1e-3))))
    for _ in range(n): run()            # warm clocks
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pad(c):
    return (c + 15) // 16 * 16


tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
print(f"{'layer':8s} {'shape':>22s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s} | lost vs 0.5 peak (ms)")
ONLY = [v for v in os.environ.get("LB_LAYERS", "").split(",") if v]
for name, cin, cout, n, s in LAYERS:
    if ONLY and name not in ONLY:
        continue
    no = (n - 1) // s + 1
    cinp, coutp = pad(cin), pad(cout)
    x = torch.randn(B, n, n, n, cinp, device=DEV).to(tdt)
    if cinp != cin:
        x[..., cin:] = 0
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cinp, coutp, dt) // 2, dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cinp, coutp, dt, stream_of()), "pack")
    y = torch.empty((B, no, no, no, cout), dtype=tdt, device=DEV)
    st = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, cout, no, no, no), dtype=torch.uint8, device=DEV)
    dy = torch.randn(B, no, no, no, cout, device=DEV).to(tdt)
    dx = torch.empty((B, n, n, n, cinp), dtype=tdt, device=DEV)
    dw = torch.empty((cout, cin, 3, 3, 3), device=DEV)
    nb = lib.dgtta_conv3d_wgrad_ws_bytes(B, cin, cout, no, no, no)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    gf = B * 2 * 27 * cin * cout * no ** 3 / 1e9
    t_f = timed(lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cinp, ptr(wpack), None, ptr(y), cout, ptr(st), B, cin, cout, cinp, coutp, n, n, n, s, dt, 0, stream_of()), "fwd"))
    t_d = timed(lambda: check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wpack), ptr(dx), cinp, B, cin, cout, cinp, coutp, n, n, n, s, 0, dt, 0, stream_of()), "dgrad")) if name != "enc0.0" else 0.0
    t_w = timed(lambda: check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cinp, ptr(dy), cout, ptr(dw), None, ptr(ws), nb, B, cin, cout, n, n, n, s, 0, dt, 0, stream_of()), "wgrad"))
    ideal = gf / 1250.0          # ms at 0.5 of 2.5 PF
    lost = [max(0.0, t - ideal) for t in (t_f, t_d, t_w)]
    for k, t in zip(("fwd", "dgrad", "wgrad"), (t_f, t_d, t_w)):
        tot[k][0] += t
        tot[k][1] += gf if t > 0 else 0.0
    tf = lambda t: (gf / t) if t > 0 else 0.0
    print(f"{name:8s} {f'{cin}->{cout} @{n}^3 s{s}':>22s} {gf:8.1f} | {t_f:8.3f} {tf(t_f):7.0f} | {t_d:8.3f} {tf(t_d):7.0f} | {t_w:8.3f} {tf(t_w):7.0f} | "
          f"{lost[0]:.3f} {lost[1]:.3f} {lost[2]:.3f}", flush=True)
    del x, y, dy, dx, dw, ws, st, wpack
    torch.cuda.empty_cache()
for k, (t, g) in tot.items():
    print(f"sum {k}: {t:.2f} ms per batch-{B} pass, {g / t:.0f} TFLOP/s average; per epoch x4 passes = {4 * t:.1f} ms")
