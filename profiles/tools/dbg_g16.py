import sys, torch
sys.path.insert(0, ".")
from dg_tta_amd import _lib, ops
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
DEV = "cuda:0"
for dtype in (torch.float16, torch.bfloat16):
    dt = ops.dtype_code(dtype)
    torch.manual_seed(5)
    B, N, C = 2, 32, 16
    both = torch.randn(2 * B, N, N, N, C, device=DEV) * 3
    v = N ** 3
    nbytes = lib.dgtta_softdice_ws_bytes(B, C, v)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dice, loss = torch.empty(B, C, device=DEV), torch.empty((), device=DEV)
    check(lib.dgtta_softdice_fwd(ptr(both[:B]), ptr(both[B:]), ptr(dice), ptr(loss), ptr(ws), nbytes, B, C, v, C, 1, 1, stream_of()), "fwd")
    g32 = torch.empty_like(both)
    g16 = torch.empty(2 * B, N, N, N, C, dtype=dtype, device=DEV)
    check(lib.dgtta_softdice_bwd(ptr(both[:B]), ptr(both[B:]), ptr(g32[:B]), ptr(g32[B:]), ptr(ws), 4096.0, None, B, C, v, C, 1, stream_of()), "bwd")
    check(lib.dgtta_softdice_bwd_t(ptr(both[:B]), ptr(both[B:]), ptr(g16[:B]), ptr(g16[B:]), ptr(ws), 4096.0, None, B, C, v, C, 1, dt, stream_of()), "bwd_t")
    torch.cuda.synchronize()
    ref = g32.to(dtype)
    ne = (g16 != ref)
    print(dtype, "mismatches", int(ne.sum()), "of", ne.numel(), "nan", int(torch.isnan(g16.float()).sum()))
    idx = ne.nonzero()[:8]
    for i in idx:
        i = tuple(i.tolist())
        print(i, float(g32[i]), float(g16[i]), float(ref[i]))
    g32b = torch.empty_like(both)
    check(lib.dgtta_softdice_bwd(ptr(both[:B]), ptr(both[B:]), ptr(g32b[:B]), ptr(g32b[B:]), ptr(ws), 4096.0, None, B, C, v, C, 1, stream_of()), "bwd")
    print("fp32 kernel run to run equal:", torch.equal(g32, g32b), "max rel diff g16 vs g32", float(((g16.float() - g32).abs() / (g32.abs() + 1e-12)).max()))
