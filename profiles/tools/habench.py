"""dgtta_seghead_window_accumulate_t alone: python3 profiles/tools/habench.py [fp32|fp16] [iters]; DGTTA_HA_ABL=1|2 ablations.
One 128^3 window of a 320^3 x 105 accumulator (origins cycle so that the lines are not cache resident), bf16 features."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ.setdefault("DGTTA_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgtta_hip_diag.so"))      # laboratory build: python -m dg_tta_amd.build --diag
import torch
from dg_tta_amd import _lib, ops
from dg_tta_amd._lib import check, ptr, stream_of
acc_dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.float32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 54
dev = torch.device("cuda:0")
lib = _lib.load()
n, P, C = 320, 128, 105
acc = torch.zeros((n, n, n, C), dtype=acc_dt, device=dev)
nsum = torch.zeros((n, n, n), device=dev)
z = torch.randn((P, P, P, 32), device=dev).to(torch.bfloat16)
w = torch.randn((C, 32), device=dev) * 0.1
b = torch.randn(C, device=dev)
g = torch.rand((P, P, P), device=dev) + 0.1
origins = [(x, y, zz) for x in (0, 96, 192) for y in (0, 96, 192) for zz in (0, 96, 192)]
def run(k):
    x0, y0, z0 = origins[k % len(origins)]
    check(lib.dgtta_seghead_window_accumulate_t(ptr(z), ptr(w), ptr(b), ptr(g), ptr(acc), ptr(nsum), 32, C, P, P, P, n, n, n, x0, y0, z0,
                                                ops.BF16, ops.F32 if acc_dt == torch.float32 else ops.F16, stream_of(dev)), "ha")
for k in range(5): run(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(iters): run(k)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gb = P ** 3 * C * acc.element_size() * 2 / 1e9 + P ** 3 * 64 / 1e9
print(f"head_accumulate {sys.argv[1] if len(sys.argv) > 1 else 'fp32'}: {ms*1e3:.1f} us per window, {gb/ms:.2f} TB/s of {gb:.2f} GB algorithmic")
