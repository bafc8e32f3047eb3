"""MIND3D forward at the bench's shape (one 128^3 sample per launch, 16-bit NDHWC output rows of 16), sustained: ms per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd.mind import MIND3D
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, 1, 128, 128, 128, device=dev)
noise = torch.randn(B, 12, 128, 128, 128, device=dev)
m = MIND3D()
def run(): return m(x, noise, out_dtype=torch.float16)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(f"MIND3D fwd B={B} 128^3 (noise given): {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call [{os.environ.get('DGTTA_MIND_CG2', 'default CG=4')}]")
