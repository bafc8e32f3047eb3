"""Where does a per-epoch loss deviation of 16-bit storage come from?  The referee's workload (64^3, pre-trained instance of the
tree) run by the product in fp32 and in a 16-bit type on the same draw stream, with the per-(step, class) soft Dice of the
consistency loss recorded: prints, for each epoch, the steps and classes whose Dice differs most and the class's mass.
usage: step_loss_ab.py [fp16|bf16] [epochs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from dg_tta_amd import ops
from dg_tta_amd.optim import HipAdamW
from dg_tta_amd.tta.model_utils import get_model_from_network
from dg_tta_amd.tta.torch_utils import release_resident
from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
from dg_tta_amd.utils import disable_internal_augmentation
from oracle.replay import cpu_rng_for_device_draws

other = sys.argv[1] if len(sys.argv) > 1 else "fp16"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
args = bench.parse_args(["--referee-epochs", str(E), "--referee-accum", "16"])
args.size = args.referee_patch
args.pretrain_hooks, args.target_noise, args.lr = "GIN_MIND", None, 1e-5
state, prep = bench.pretrained_weights(args, dev)
P, A, seed = [args.referee_patch] * 3, 16, 6006
rec = {}
orig = ops.consistency_loss


def run(dtype):
    log = []

    def spy(ta, tb, start):
        loss, dice = orig(ta, tb, start)
        log.append((dice.detach().float().cpu().clone(), ta.detach().float().sum((0, 2, 3, 4)).cpu() if ta.dim() == 5 else None))
        return loss, dice
    ops.consistency_loss = spy
    import dg_tta_amd.tta.tta as T
    T.ops.consistency_loss = spy
    net, cfg, mapping, modmod, data = bench.build_workload(args, dev, 0, dtype)
    cfg.update(epochs=E, patches_to_be_accumulated=A)
    names = cfg["optimized_labels"]
    model = get_model_from_network(net, modmod, None)
    fused = _fuse_head_if_possible(model, modmod, mapping, names)
    opt = HipAdamW(model.parameters(), lr=cfg["lr"], grad_scale=model.loss_scale)
    disable_internal_augmentation()
    release_resident()
    with cpu_rng_for_device_draws():
        torch.manual_seed(seed)
        np.random.seed(seed)
        losses, dices = tta_unit(model, opt, cfg, [data], P, mapping, modmod, dev, fused)
    release_resident()
    return losses, log


l32, g32 = run("fp32")
l16, g16 = run(other)
print("per-epoch loss fp32", [round(float(x), 6) for x in l32])
print(f"per-epoch loss {other}", [round(float(x), 6) for x in l16], "delta", [f"{abs(float(a) - float(b)):.2e}" for a, b in zip(l32, l16)])
print("calls to the loss:", len(g32), "dice shape", tuple(g32[0][0].shape))
for i, ((d32, m32), (d16, _)) in enumerate(zip(g32, g16)):
    dd = (d32 - d16).abs()
    if float(dd.max()) > 2e-3:
        flat = dd.flatten()
        top = flat.topk(min(4, flat.numel()))
        rows = []
        for v, idx in zip(top.values, top.indices):
            pos = np.unravel_index(int(idx), dd.shape)
            rows.append(f"{pos}: fp32 {float(d32[pos]):.5f} {other} {float(d16[pos]):.5f}")
        print(f"loss call {i}: max |dice delta| {float(dd.max()):.3e}; mean over entries {float(dd.mean()):.3e}; worst " + " | ".join(rows))
