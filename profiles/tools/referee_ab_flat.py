"""Which side of a referee run does a kernel switch move: the pre-trained instance or the TTA run on it?  Pre-training with
DGTTA_<switch>=argv[2], the engine's TTA runs with =argv[3]; 2 epochs x 16 accumulation steps at 64^3 (epoch 1 = after the
first optimizer step).  usage: referee_ab_flat.py WGRAD_FLAT 1 0"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from dg_tta_amd import _lib
lib = _lib.load()
name, pre, tta = "DGTTA_" + sys.argv[1], sys.argv[2], sys.argv[3]
epochs = sys.argv[4] if len(sys.argv) > 4 else "2"
orig = bench.pretrained_weights


def patched(args, device):
    os.environ[name] = pre
    lib.dgtta_reload_env()
    r = orig(args, device)
    os.environ[name] = tta
    lib.dgtta_reload_env()
    return r


bench.pretrained_weights = patched
args = bench.parse_args(["--referee-epochs", epochs, "--referee-accum", "16"])
out = bench.referee_tta_run(args, torch.device("cuda:0"))
print("\n" + json.dumps({"skipped": {k: out[k]["skipped_optimizer_steps"] for k in ("fp16",)}, "switch": name, "pretraining": pre, "tta": tta, "oracle_loss": out["oracle"]["loss_per_epoch"],
                         **{k: {"loss_per_epoch": out[k]["loss_per_epoch"], "pseudo_dice": out[k]["pseudo_dice"], "hard_dice": out[k]["hard_dice"]} for k in ("fp32", "fp16", "bf16")}}))
