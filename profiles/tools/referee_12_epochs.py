"""BASELINE config 2's schedule (12 TTA epochs x 16 accumulation steps, GIN + affine in both branches, MIND net, AdamW) refereed by
the CPU ORACLE end to end: bench.referee_tta_run at --referee-epochs 12 --referee-accum 16 (64^3 patches of an 80^3 target-domain
volume, weights pre-trained by the bench's recipe at that size), the product in fp32 / fp16 / bf16 storage on the same draw stream.
~5 min of host time for the oracle.  usage: referee_12_epochs.py [lr] > profiles/r06_dice_delta_12_epochs.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
lr = sys.argv[1] if len(sys.argv) > 1 else "3e-4"
bench.REFEREE_LR = float(lr)          # (the referee run pins its own recipe: GIN + MIND pre-training, standard target, this rate)
args = bench.parse_args(["--referee-epochs", "12", "--referee-accum", "16"])
out = bench.referee_tta_run(args, torch.device("cuda:0"))
sys.stdout.write("\n" + json.dumps(out, indent=1) + "\n")      # (tta_unit prints its own progress lines before this)
