import sys, ctypes as C, torch
sys.path.insert(0, '.')
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
DEV = "cuda:0"
torch.manual_seed(0)
B, Cin, Cout, n = int(sys.argv[4]) if len(sys.argv) > 4 else 2, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt, tdt = 2, torch.float16
w = torch.randn(Cout, Cin, 3, 3, 3, device=DEV) * 0.05
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(Cin, Cout, dt) // 2, dtype=tdt, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), Cin, Cout, Cin, Cout, dt, stream_of()), "pack")
dy = torch.randn(B, n, n, n, Cout, device=DEV).to(tdt)
yprev = (torch.randn(B, n, n, n, Cin, device=DEV) * 2 + 0.3).to(tdt)
mr = torch.stack([yprev.float().mean((1, 2, 3)), 1.0 / (yprev.float().var((1, 2, 3), unbiased=False) + 1e-5).sqrt()], -1).contiguous()
gamma = torch.rand(Cin, device=DEV) + 0.5
beta = torch.randn(Cin, device=DEV) * 0.3
gin = torch.empty(B, n, n, n, Cin, device=DEV, dtype=tdt)
gb = lib.dgtta_conv3d_stats_bytes(B, Cin, n, n, n)
gs = torch.zeros(gb // 8 + 1, dtype=torch.float64, device=DEV)
prod = C.c_int(0)
check(lib.dgtta_conv3d_k3_dgrad_gstats(ptr(dy), Cout, ptr(wpack), ptr(gin), Cin, B, Cin, Cout, Cin, Cout, n, n, n, ptr(yprev), Cin, ptr(mr), ptr(gamma), ptr(beta), 0.01, ptr(gs), gb, C.byref(prod), dt, 0, stream_of()), "dg")
torch.cuda.synchronize()
print("produced", prod.value, "header", gs[:1].view(torch.int64).item())
nblk = gs[:1].view(torch.int64).item()
part = gs[32:32 + B * nblk * Cin * 2].view(B, nblk, Cin, 2).sum(1)
g = gin.float(); y = yprev.float()
a = (y - mr[:, None, None, None, :, 0]) * mr[:, None, None, None, :, 1] * gamma + beta
gp = torch.where(a > 0, g, g * 0.01)
ref0 = gp.sum((1, 2, 3)).double(); ref1 = (gp * y).sum((1, 2, 3)).double()
print("sum g'   max rel err", float(((part[..., 0] - ref0).abs() / ref0.abs().clamp_min(1)).max()))
print("sum g'y  max rel err", float(((part[..., 1] - ref1).abs() / ref1.abs().clamp_min(1)).max()))
print(part[0, :4], ref0[0, :4], ref1[0, :4])
# timing: plain data gradient vs the fused form (same launch shape)
def t(fn, k=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
plain = lambda: check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), Cout, ptr(wpack), ptr(gin), Cin, B, Cin, Cout, Cin, Cout, n, n, n, 1, 0, dt, 0, stream_of()), "d")
fused = lambda: check(lib.dgtta_conv3d_k3_dgrad_gstats(ptr(dy), Cout, ptr(wpack), ptr(gin), Cin, B, Cin, Cout, Cin, Cout, n, n, n, ptr(yprev), Cin, ptr(mr), ptr(gamma), ptr(beta), 0.01, ptr(gs), gb, C.byref(prod), dt, 0, stream_of()), "dg")
dyo = torch.empty_like(gin); dg = torch.empty(Cin, device=DEV); db = torch.empty(Cin, device=DEV)
nb = lib.dgtta_instnorm_ws_bytes(B, Cin, n ** 3); ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
inb = lambda: check(lib.dgtta_instnorm_lrelu_bwd(ptr(gin), Cin, ptr(yprev), Cin, ptr(gamma), ptr(beta), ptr(mr), ptr(dyo), Cin, ptr(dg), ptr(db), ptr(ws), nb, B, Cin, n ** 3, 0.01, 0, dt, stream_of()), "inb")
ing = lambda: check(lib.dgtta_instnorm_lrelu_bwd_gstats(ptr(gin), Cin, ptr(yprev), Cin, ptr(gamma), ptr(beta), ptr(mr), ptr(dyo), Cin, ptr(dg), ptr(db), ptr(gs), ptr(ws), nb, B, Cin, n ** 3, 0.01, 0, dt, stream_of()), "ing")
print(f"B={B} {Cin}->{Cout} {n}^3: dgrad plain {t(plain):.3f} ms, fused {t(fused):.3f} ms; IN-bwd with reduction pass {t(inb):.3f} ms, from gstats {t(ing):.3f} ms")
