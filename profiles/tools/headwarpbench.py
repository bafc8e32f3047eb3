"""Micro-bench at the bench's launch shape (8 x 128^3, 32 -> 16): head then warp (two launches each way) vs the fused pair."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib, ops
from dg_tta_amd._lib import check, ptr, stream_of
from dg_tta_amd.tta.augmentation_utils import get_rand_affine
lib = _lib.load()
DEV = "cuda:0"
B, N, CIN, NS = 8, 128, 32, 16
dt, tdt = (2, torch.float16) if os.environ.get("HW_DT", "fp16") == "fp16" else (1, torch.bfloat16)
torch.manual_seed(0)
z = torch.randn(B, N, N, N, CIN, device=DEV).to(tdt)
w = torch.randn(105, CIN, device=DEV) * 0.1
bias = torch.randn(105, device=DEV)
sel = (torch.arange(NS) * 3).to(torch.int32).to(DEV)
_, rinv = get_rand_affine(B)
rinv = rinv.float().contiguous()
th = rinv.to(DEV)
out = torch.empty(B, N, N, N, NS, device=DEV)
gout = torch.randn(B, N, N, N, NS, device=DEV)
gz = torch.empty(B, N, N, N, CIN, device=DEV, dtype=tdt)
dws, dbs = torch.empty(NS, CIN, device=DEV), torch.empty(NS, device=DEV)
V = N ** 3
nb_f = lib.dgtta_seghead_warp_bwd_ws_bytes(B, CIN, NS, N, N, N)
nb_u = lib.dgtta_seghead_bwd_ws_bytes(B, CIN, NS, V)
ws = torch.empty(max(nb_f, nb_u), dtype=torch.uint8, device=DEV)
logits = torch.empty(B, N, N, N, NS, device=DEV)
glog = torch.empty(B, N, N, N, NS, device=DEV)
st = stream_of()
def fwd_unfused():
    check(lib.dgtta_seghead_fwd(ptr(z), CIN, ptr(w), ptr(bias), ptr(sel), NS, ptr(logits), 1, NS, B, CIN, V, dt, st), "head")
    check(lib.dgtta_affine_warp3d_fwd(ptr(logits), ptr(th), ptr(out), B, NS, N, N, N, N, N, N, 1, NS, NS, 0, 0, 1, None, st), "warp")
def fwd_fused():
    check(lib.dgtta_seghead_warp_fwd(ptr(z), ptr(w), ptr(bias), ptr(sel), NS, ptr(th), ptr(out), B, CIN, N, N, N, 1, dt, st), "fw")
def bwd_unfused():
    check(lib.dgtta_affine_warp3d_bwd(ptr(gout), ptr(th), ptr(glog), B, NS, N, N, N, N, N, N, 1, NS, NS, 0, 1, st), "wb")
    check(lib.dgtta_seghead_bwd(ptr(z), CIN, ptr(glog), NS, ptr(w), ptr(sel), NS, ptr(gz), CIN, ptr(dws), ptr(dbs), ptr(ws), nb_u, B, CIN, V, 0, dt, st), "hb")
def bwd_fused():
    check(lib.dgtta_seghead_warp_bwd(ptr(z), ptr(gout), ptr(th), ptr(rinv), ptr(w), ptr(sel), NS, ptr(gz), ptr(dws), ptr(dbs), ptr(ws), nb_f, B, CIN, N, N, N, 1, 0, dt, st), "fb")
def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, fn in (("fwd unfused", fwd_unfused), ("fwd fused", fwd_fused), ("bwd unfused (incl. head wgrad + bias)", bwd_unfused), ("bwd fused (incl. head wgrad + bias)", bwd_fused)):
    print(f"{name:42s} {t(fn):.3f} ms")

# round 6: the head products on the fp32 matrix cores (DGTTA_HEADWARP_MFMA, default on) and the logit gradient in 16 bits
g16 = gout.to(tdt)
def bwd_fused_g16():
    check(lib.dgtta_seghead_warp_bwd_g16(ptr(z), ptr(g16), ptr(th), ptr(rinv), ptr(w), ptr(sel), NS, ptr(gz), ptr(dws), ptr(dbs), ptr(ws), nb_f, B, CIN, N, N, N, 1, 0, dt, st), "fb16")
for mf in ("1", "0"):
    os.environ["DGTTA_HEADWARP_MFMA"] = mf
    lib.dgtta_reload_env()
    for name, fn in (("fwd fused", fwd_fused), ("bwd fused, fp32 gradient", bwd_fused), ("bwd fused, 16-bit gradient", bwd_fused_g16)):
        t(fn, 40)                       # (sustained clocks: a short burst reads faster than the kernel inside an epoch)
        print(f"DGTTA_HEADWARP_MFMA={mf} {name:30s} {t(fn, 40):.3f} ms")
