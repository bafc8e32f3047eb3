"""In-kernel clock, barrier wait and MFMA-busy share of the weight-gradient ring kernel (DGTTA_WGRAD_RING_CLK=6, diagnostic build, fp16, 8 x 128^3 x 32 -> 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DGTTA_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgtta_hip_diag.so"))      # laboratory build: python -m dg_tta_amd.build --diag
os.environ["DGTTA_WGRAD_RING_CLK"] = sys.argv[1] if len(sys.argv) > 1 else "6"
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
B, n, c = 8, 128, 32
DEV = "cuda:0"
x = torch.randn(B, n, n, n, c, device=DEV).half()
dy = torch.randn(B, n, n, n, c, device=DEV).half()
dw = torch.empty((c, c, 3, 3, 3), device=DEV)
nb = lib.dgtta_conv3d_wgrad_ws_bytes(B, c, c, n, n, n)
ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
run = lambda: check(lib.dgtta_conv3d_k3_wgrad(ptr(x), c, ptr(dy), c, ptr(dw), None, ptr(ws), nb, B, c, c, n, n, n, 1, 0, 2, 2, stream_of()), "wgrad")
for _ in range(300):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
G = 256
# the slabs start behind the bias partials of the workspace (size not exported): find the stamp block by its pattern
f = ws.view(torch.float32)
cand = ((f[1:-7:1] > 3e4) & (f[1:-7:1] < 5e5) & (f[5:-3:1] > 3e4) & (f[5:-3:1] < 5e5) & (f[3:-5:1] == 0) & (f[7:-1:1] == 0)).nonzero()
start = int(cand[0]) if len(cand) else G * 27 * 1024
o = f[start: start + G * 8 * 4].reshape(G, 8, 4).cpu()
cyc, rt, wait = o[..., 0], o[..., 1], o[..., 2]
clk = (cyc / rt * 100e6).flatten()
mfma_cycles = 2 * 128 * 2 * 56 * 32          # per SIMD: 2 waves x (2 jobs x 128 slices) x 56 MFMAs (32x32x16) x 32 cycles
print(f"launch {e0.elapsed_time(e1)*1e3:.1f} us incl. reduce; cycles per wave mean {float(cyc.mean()):.0f}; clock median {float(clk.median())/1e9:.3f} GHz")
print(f"barrier + DMA wait per wave: mean {100*float((wait/cyc).mean()):.1f} % (by wave: " + " ".join(f"{100*float(v):.0f}" for v in (wait/cyc).mean(0)) + ")")
print(f"MFMA-busy share of the wave time: {100*mfma_cycles/float(cyc.mean()):.1f} %")
