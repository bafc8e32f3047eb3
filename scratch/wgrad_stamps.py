"""Diagnostic: per-segment cycle totals of the MFMA wgrad kernel (DGTTA_WGRAD_ABL=6 build path)."""
import os, sys, torch
os.environ["DGTTA_WGRAD_ABL"] = "6"
sys.path.insert(0, '.')
from dg_tta_amd import _lib
from dg_tta_amd._lib import check, ptr, stream_of
lib = _lib.load()
cin, cout, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
DEV = "cuda:0"
x = torch.randn(1, n, n, n, cin, device=DEV).bfloat16()
dy = torch.randn(1, n, n, n, cout, device=DEV).bfloat16()
dw = torch.empty((cout, cin, 3, 3, 3), device=DEV)
nb = lib.dgtta_conv3d_wgrad_ws_bytes(1, cin, cout, n, n, n)
ws = torch.zeros(nb // 4, dtype=torch.float32, device=DEV)
for _ in range(2):
    check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cin, ptr(dy), cout, ptr(dw), None, ptr(ws), nb * 1, 1, cin, cout, n, n, n, 1, 0, 1, 2, stream_of()), "wgrad")
torch.cuda.synchronize()
BASE = 13312   # floats ahead of the slabs in the workspace (bias partials)
nsl = (ws.numel() - BASE) // (27 * 1024)
sl = ws[BASE:BASE + nsl * 27 * 1024].view(-1, 27 * 1024).cpu()
used = sl[:, -64:-32].reshape(-1, 4, 8)
used = used[used[:, 0, 2] > 0]
names = ["prologue", "load_issue", "mfma_loop", "wait+transpose+lds_write", "barrier", "slab_write"]
print("workgroups with stamps:", used.shape[0], " total cycles/wave:", float(used[:, :, :6].sum(-1).mean()))
for k, nm in enumerate(names):
    print(f"  {nm:26s} mean {float(used[:, :, k].mean()):10.0f}  max {float(used[:, :, k].max()):10.0f}")
