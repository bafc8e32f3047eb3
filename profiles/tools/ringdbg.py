"""Debug aid for the D-ring conv kernel: runs DGTTA_CONV_RING=1 against the generic MFMA kernel on one case and prints where they differ.
usage: ringdbg.py B cin cout D H W [fp16|bf16] [dgrad]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
B, cin, cout, D, H, W = [int(a) for a in sys.argv[1:7]]
dts = sys.argv[7] if len(sys.argv) > 7 else "fp16"
dgrad = len(sys.argv) > 8 and sys.argv[8] == "dgrad"
dt = {"bf16": 1, "fp16": 2}[dts]
tdt = {1: torch.bfloat16, 2: torch.float16}[dt]
DEV = "cuda:0"
torch.manual_seed(0)
w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5)
bias = None if dgrad else torch.randn(cout, device=DEV)
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // 2, dtype=tdt, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
ci, co = (cout, cin) if dgrad else (cin, cout)
x = torch.randn(B, D, H, W, ci, device=DEV).to(tdt)

def run(ring):
    os.environ["DGTTA_CONV_RING"] = ring
    os.environ["DGTTA_CONV_ROWS"] = "0"
    lib.dgtta_reload_env()
    y = torch.full((B, D, H, W, co), float("nan"), dtype=tdt, device=DEV)
    st = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, co, D, H, W), dtype=torch.uint8, device=DEV)
    if dgrad:
        check(lib.dgtta_conv3d_k3_dgrad(ptr(x), ci, ptr(wpack), ptr(y), co, B, cin, cout, cin, cout, D, H, W, 1, 0, dt, 2, stream_of()), "dgrad")
    else:
        check(lib.dgtta_conv3d_k3_fwd(ptr(x), ci, ptr(wpack), ptr(bias), ptr(y), co, ptr(st), B, cin, cout, cin, cout, D, H, W, 1, dt, 2, stream_of()), "fwd")
    torch.cuda.synchronize()
    hdr = int(st.view(torch.int64)[0])
    sums = st[256:].view(torch.float64)[: B * max(hdr, 1) * co * 2].reshape(B, max(hdr, 1), co, 2).sum(1) if hdr > 0 else None
    return y.float(), sums

y0, s0 = run("0")
y1, s1 = run("1")
err = (y1 - y0).abs()
bad = ~(err < 0.02 * y0.abs().max() + 1e-3)
print(f"max |y| {float(y0.abs().max()):.3f}  max err {float(torch.nan_to_num(err, nan=1e9).max()):.4g}  wrong {int(bad.sum())} of {bad.numel()}  nan {int(torch.isnan(y1).sum())}")
if bad.any():
    for name, dims in (("b", (1, 2, 3, 4)), ("d", (0, 2, 3, 4)), ("h", (0, 1, 3, 4)), ("w", (0, 1, 2, 4)), ("c", (0, 1, 2, 3))):
        fr = bad.float().mean(dims)
        print(f"  wrong fraction by {name}:", " ".join(f"{float(f):.2f}" for f in fr))
    idx = bad.nonzero()[:8]
    for i in idx:
        i = tuple(int(v) for v in i)
        print("   ", i, float(y1[i]), "ref", float(y0[i]))
if s0 is not None and s1 is not None:
    print("stats: sum rel err", float(((s1 - s0).abs() / (s0.abs() + 1e-3)).max()))
