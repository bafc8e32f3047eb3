"""Micro-bench of the consistency loss at the bench's launch shape (4 pairs x 128^3 x 16 fp32): forward, backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
both = (torch.randn(8, 128, 128, 128, 16, device=DEV) * 2).permute(0, 4, 1, 2, 3).requires_grad_(True)
def run():
    a, b = both[:4], both[4:]
    a._dgtta_pair = b._dgtta_pair = both
    loss, _ = ops.consistency_loss(a, b)
    return loss
t = {"fwd": 0.0, "bwd": 0.0}
for it in range(12):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); loss = run(); e[1].record(); loss.backward(); e[2].record(); torch.cuda.synchronize(); both.grad = None
    if it >= 2:
        t["fwd"] += e[0].elapsed_time(e[1]) / 10; t["bwd"] += e[1].elapsed_time(e[2]) / 10
print(f"SOFTDICE16={os.environ.get('DGTTA_SOFTDICE16', '-')}: fwd {t['fwd']:.3f} ms ({1.074 / t['fwd']:.2f} TB/s), bwd {t['bwd']:.3f} ms ({2.147 / t['bwd']:.2f} TB/s)")
