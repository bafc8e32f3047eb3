"""Per-segment cycle stamps and in-kernel clock of the D-ring conv kernel (DGTTA_RING_ABL=6, fp16, 128^3 32->32, batch 8).
usage: ring_stamps.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DGTTA_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgtta_hip_diag.so"))      # laboratory build: python -m dg_tta_amd.build --diag
os.environ["DGTTA_RING_ABL"] = "6"
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, cin, cout, dt = 128, 32, 32, 2
DEV = "cuda:0"
x = torch.randn(B, n, n, n, cin, device=DEV).half()
w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // 2, dtype=torch.float16, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
y = torch.empty((B, n, n, n, cout), dtype=torch.float16, device=DEV)
nbytes = max(lib.dgtta_conv3d_stats_bytes(B, cout, n, n, n), (1 << 23) + 256 * 8 * 8 * 8 + 4096)
st = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
run = lambda: check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, ptr(st), B, cin, cout, cin, cout, n, n, n, 1, dt, 2, stream_of()), "fwd")
for _ in range(200):      # ~0.2 s of back-to-back launches before the stamped one counts (clock settles under load)
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
o = st.view(torch.float64)[(1 << 20):(1 << 20) + 256 * 8 * 8].reshape(256, 8, 8).cpu()
names = ["prologue", "rows 0-6", "wait+barrier", "rows 7-15", "epilogue"]
tot = o[..., 5]
print(f"launch {e0.elapsed_time(e1)*1e3:.1f} us (stamped build); cycles per wave: mean {float(tot.mean()):.0f} min {float(tot.min()):.0f} max {float(tot.max()):.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {float(o[..., i].mean()):10.0f}  ({100*float(o[..., i].mean()/tot.mean()):5.1f} %)  min {float(o[..., i].min()):10.0f} max {float(o[..., i].max()):10.0f}")
wt = o[..., 2].mean(0) / 128
print("wait+barrier per step by wave:", " ".join(f"{float(x):.0f}" for x in wt), "| epilogue:", " ".join(f"{float(x):.0f}" for x in o[..., 4].mean(0) / 128))
clk = (o[..., 5] / o[..., 6] * 100e6).flatten()
print(f"in-kernel clock: median {float(clk.median())/1e9:.3f} GHz (min {float(clk.min())/1e9:.3f}, max {float(clk.max())/1e9:.3f}); kernel span {float((o[..., 6].max())/100):.1f} us")
