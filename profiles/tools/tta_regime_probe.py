"""After bench.py's in-bench pre-training: is the reference's mask (sum_c logits > 0, tta.py:263-265) alive on the target case,
and what do a few adaptation epochs do to the loss and to the hard Dice vs ground truth?  usage: tta_regime_probe.py [epochs] [lr] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lr = sys.argv[2] if len(sys.argv) > 2 else "3e-4"
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp16"
args = bench.parse_args(["--lr", lr] + sys.argv[4:])
dev = torch.device("cuda:0")
state, rep = bench.pretrained_weights(args, dev)
print({k: v for k, v in rep.items() if k not in ("task", "recipe")}, flush=True)
torch.manual_seed(4321); np.random.seed(4321)
r = bench.EpochRunner(args, dev, 0, dtype)
from dg_tta_amd.tta.torch_utils import get_batch
with torch.no_grad():
    r.model.eval()
    imgs, labels = get_batch(r.data, [0], r.patch, "center", dev)
    out = r.model(imgs[0]).float()
    r.model.train()
s = out.sum(1)
print(f"selected-class logits on the centre patch: sum > 0 on {float((s > 0).float().mean()):.4f} of the voxels; range [{float(out.min()):.2f}, {float(out.max()):.2f}]; "
      f"foreground voxels {float((labels[0] > 0).float().mean()):.4f}, of which sum > 0: {float((s[labels[0][:, 0] > 0] > 0).float().mean()):.4f}", flush=True)
_, before = r.final_labels()
print(f"hard Dice vs GT before: {float(before.nanmean()):.4f}", flush=True)
for e in range(epochs):
    r.epoch()
    _, pc = r.final_labels()
    print(f"epoch {e}: loss {r.losses[-1]:.5f}, pseudo-Dice {r.dices[-1]:.4f}, hard Dice vs GT {float(pc.nanmean()):.4f}", flush=True)
