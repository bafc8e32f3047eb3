"""Diagnostic: per-segment cycle totals of the row-reuse conv kernel (DGTTA_ROWS_ABL=6 build path)."""
import os, sys, torch
os.environ["DGTTA_ROWS_ABL"] = "6"
sys.path.insert(0, '.')
os.environ.setdefault("DGTTA_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgtta_hip_diag.so"))      # laboratory build: python -m dg_tta_amd.build --diag
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
cin, cout, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
DEV = "cuda:0"
x = torch.randn(1, n, n, n, cin, device=DEV).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, 1) // 2, dtype=torch.bfloat16, device=DEV)
check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, 1, stream_of()), "pack")
y = torch.empty((1, n, n, n, cout), dtype=torch.bfloat16, device=DEV)
nb = max(lib.dgtta_conv3d_stats_bytes(1, cout, n, n, n), (4096 + 256 * 8 * 10 + 64) * 8)
st = torch.zeros(nb // 8 + 1, dtype=torch.float64, device=DEV)
for _ in range(3):
    check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, ptr(st), 1, cin, cout, cin, cout, n, n, n, 1, 1, 2, stream_of()), "fwd")
torch.cuda.synchronize()
t = st[4096:4096 + 256 * 8 * 8].view(256, 8, 8).cpu()
names = ["dma_wait", "bar+Bfrag+bar", "dma_issue", "mfma_loop", "epi_rest(stats)", "epi_barrier", "epi_cvt+slab", "epi_stores"]
tot = t.sum(-1)
print("cycles per wave (mean over waves): total", float(tot.mean()))
for k, nm in enumerate(names):
    print(f"  {nm:14s} mean {float(t[:, :, k].mean()):10.0f}  min {float(t[:, :, k].min()):10.0f}  max {float(t[:, :, k].max()):10.0f}")
clk = st[4096 + 256 * 8 * 8:4096 + 256 * 8 * 10].view(256 * 8, 2).cpu()
ghz = (clk[:, 0] / clk[:, 1] * 0.1).median()
print(f"in-kernel clock (median over waves): {float(ghz):.3f} GHz; kernel span {float(clk[:, 1].median()) / 100:.1f} us")
