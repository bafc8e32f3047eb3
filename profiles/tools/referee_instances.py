"""How much of a 16-bit run's deviation from the oracle is the pre-trained instance?  The 3-epoch x 8-step referee run of
tests/test_gpu_referee.py (bench.referee_tta_run) over several pre-training seeds (bench.PRETRAIN["seed"]: the patch / augmentation
draws of the 550 pre-training steps; seed 5 is the bench's): per instance the engine's fp32 / fp16 / bf16 against the CPU oracle.
usage: referee_instances.py 5 11 12 13 14 15 > gpurun_out/r06_referee_instances.json
(RI_EPOCHS / RI_ACCUM / RI_LR: another schedule, e.g. 12 / 16 / 1e-5 = BASELINE config 2 at the plan's rate)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
seeds = [int(a) for a in sys.argv[1:]] or [5, 11, 12]
rows = []
for sd in seeds:
    bench.PRETRAIN["seed"] = sd
    bench._PRETRAINED.clear()
    if os.environ.get("RI_LR"):
        bench.REFEREE_LR = float(os.environ["RI_LR"])
    args = bench.parse_args(["--referee-patch", "64", "--referee-epochs", os.environ.get("RI_EPOCHS", "3"), "--referee-accum", os.environ.get("RI_ACCUM", "8")])
    out = bench.referee_tta_run(args, torch.device("cuda:0"))
    row = {"pretrain_seed": sd, "epochs": out["epochs"], "accum": out["accum"], "lr": out["lr"], "oracle_loss_per_epoch": out["oracle"]["loss_per_epoch"],
           "oracle_hard_dice": [out["oracle"]["hard_dice_vs_gt_before"], out["oracle"]["hard_dice_vs_gt_after"]],
           "mask_voxel_fraction": out["pretraining"]["voxels_with_positive_mapped_logit_sum"]}
    for k in ("fp32", "fp16", "bf16"):
        e = out[k]
        row[k] = {"loss": e["loss"], "loss_per_epoch": e["loss_per_epoch"], "pseudo_dice": e["pseudo_dice"], "hard_dice": e["hard_dice"],
                  "labels_where_margin_gt_1e-3": e["label_agreement_where_margin_gt_1e-3"], "loss_tolerance": e["loss_tolerance"],
                  "within_tolerance": e["within_tolerance"], "dice_within_tolerance": e["dice_within_tolerance"]}
    rows.append(row)
    print(f"seed {sd}: " + "; ".join(f"{k} loss {row[k]['loss']:.2e} pseudo {row[k]['pseudo_dice']:.1e} hard {row[k]['hard_dice']:.1e} within {row[k]['within_tolerance']}" for k in ("fp32", "fp16", "bf16")), file=sys.stderr, flush=True)
summary = {k: {"within": sum(r[k]["within_tolerance"] for r in rows), "dice_within": sum(r[k]["dice_within_tolerance"] for r in rows), "of": len(rows),
               "loss_max": max(r[k]["loss"] for r in rows), "loss_median": sorted(r[k]["loss"] for r in rows)[len(rows) // 2]} for k in ("fp32", "fp16", "bf16")}
sys.stdout.write("\n" + json.dumps({"what": __doc__.split("usage")[0].strip(), "instances": rows, "summary": summary}, indent=1) + "\n")
