#!/usr/bin/env python3
"""Benchmark of the DG-TTA hot path on MI355X: TTA epochs per second on a 128^3 patch (BASELINE.json metric).

One "step" = one TTA epoch of the reference's inner loop (dg_tta/tta/tta.py:190-338): 16 accumulation steps x
{get_batch, 2 augmented branches (GIN -> affine warp -> MIND -> nnUNet 3d_fullres fwd -> inverse warp), masked
soft-Dice loss, backward through both branches}, one AdamW step, one centre-patch eval forward.
N > 1: one independent TTA instance per GPU (different sample per rank, no data-path collective); torch.distributed is
used only for the barrier and the max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work of one PlainConvUNet forward at 128^3
FWD_GFLOP_128 = 998.84


def conv_flops(cin, cout, vout):
    return 2.0 * 27 * cin * cout * vout


def build_workload(args, device, rank):
    from dg_tta_amd.gin import gin_hook
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.synthetic import he_init_, synthetic_case, synthetic_label_mapping
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions, TEMPLATE_PLAN
    from dg_tta_amd.unet import HipPlainConvUNet
    act = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    net = he_init_(HipPlainConvUNet(act_dtype=act, conv_impl=args.impl), seed=7)
    net.exact_zero_bias_grad = True
    net.accumulate_grads_in_place = True
    net.register_forward_pre_hook(gin_hook)
    net.register_forward_pre_hook(mind_hook)
    net = net.to(device)
    k = args.copt - 1
    mapping, names = synthetic_label_mapping(k)
    vol = args.size + 32
    data = synthetic_case(size=vol, k=k, seed=20240704 + rank)
    cfg = dict(TEMPLATE_PLAN)
    cfg.update(do_intensity_aug_in="both", do_spatial_aug_in="both", patches_to_be_accumulated=args.accum,
               optimized_labels=names, epochs=10 ** 6, ensemble_count=1, lr=1e-5)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    return net, cfg, mapping, modmod, data


class EpochRunner:
    """Runs TTA epochs back to back on one sample (the body of tta_unit, one epoch per call)."""

    def __init__(self, args, device, rank):
        from dg_tta_amd.optim import HipAdamW
        from dg_tta_amd.tta.model_utils import get_model_from_network
        from dg_tta_amd.tta.tta import _fuse_head_if_possible
        from dg_tta_amd.tta.torch_utils import fix_all, release_all
        from dg_tta_amd.utils import disable_internal_augmentation
        self.args, self.device = args, device
        net, self.cfg, self.mapping, self.modmod, data = build_workload(args, device, rank)
        self.data = [data]
        self.patch = [args.size] * 3
        self.model = get_model_from_network(net, self.modmod, None)
        self.fused = _fuse_head_if_possible(self.model, self.modmod, self.mapping, self.cfg["optimized_labels"])
        self.opt = HipAdamW(self.model.parameters(), lr=self.cfg["lr"], grad_scale=self.model.loss_scale)
        disable_internal_augmentation()
        self.model.apply(fix_all)
        self.model.apply(release_all)            # measured epochs are adaptation epochs (epoch >= start_tta_at_epoch)
        self.losses = []

    def epoch(self):
        """One adaptation epoch through the PRODUCT's own epoch function (dg_tta_amd.tta.tta.tta_epoch, the body of
        tta_unit): nothing of the loop is restated here."""
        from dg_tta_amd.tta.tta import tta_epoch
        loss, self.dice = tta_epoch(self.model, self.opt, self.cfg, self.data, self.patch, self.mapping, self.modmod,
                                    self.device, self.fused, adapt=True)
        self.losses.append(loss)


def cpu_baseline(args):
    """Times the CPU oracle (restatement of the reference, kind 'port') on a bounded sample of the same workload, as
    BASELINE.md §4 prescribes: ONE warm-up + ONE measured accumulation step (2 branches fwd + loss + bwd) on a
    `cpu_size`^3 patch (default: the full 128^3), scaled by the voxel count if smaller and by (accum + eval forward)
    to one epoch."""
    from oracle import tta as otta, unet as ounet
    n = args.cpu_size
    cores = min(len(os.sched_getaffinity(0)), 16)     # a one-GPU box's CPU share is 16 cores (oversubscribing 256 hurts)
    torch.set_num_threads(cores)
    om = ounet.init_he(ounet.PlainConvUNetOracle(), 7)
    sel = torch.arange(args.copt) * 3
    torch.manual_seed(0)
    imgs = torch.randn(1, 1, n, n, n)

    def draws(seed):
        torch.manual_seed(seed)
        return otta.draw_branch(1, [n, n, n])
    times = []
    for rep in range(1 + max(args.cpu_warmup, 0)):
        t0 = time.perf_counter()
        otta.tta_step(om, imgs, sel, draws(1 + 2 * rep), draws(2 + 2 * rep), accum=args.accum, backward=True)
        times.append(time.perf_counter() - t0)
    dt = times[-1]
    scale = (args.size / n) ** 3
    epoch_s = dt * scale * (args.accum + 1.0 / 6.0)       # eval forward ~ 1/6 of a step (1 of 6 network passes)
    return {"value": 1.0 / epoch_s, "unit": "TTA-epochs/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_warmup} warm-up + 1 measured accumulation step (2 branches fwd + loss + bwd) of the CPU "
                      f"oracle on a {n}^3 patch = {dt:.1f} s (warm-up {times[0]:.1f} s), scaled x{scale:.2f} (voxels) "
                      f"x{args.accum + 1 / 6:.2f} (steps per epoch)"}


def inference_leg(args, device):
    """BASELINE config 3's caller-side step: Gaussian sliding-window inference of ONE ensemble member
    over an `inference_size`^3 volume with 128^3 windows at step 0.5, all 105 classes accumulated in fp32, then argmax.
    Returns ms per window (network forward + accumulate) and the totals."""
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.inference import predict_sliding_window_return_logits
    from dg_tta_amd.unet import HipPlainConvUNet
    from dg_tta_amd import ops
    n = args.inference_size
    net = he_init_(HipPlainConvUNet(act_dtype={"fp32": torch.float32, "bf16": torch.bfloat16,
                                               "fp16": torch.float16}[args.dtype]), seed=7)
    net.register_forward_pre_hook(mind_hook)
    net = net.to(device)
    vol = torch.randn(1, n, n, n, generator=torch.Generator().manual_seed(3)).to(device)
    patch = [args.size] * 3
    predict_sliding_window_return_logits(net, vol[:, :args.size, :args.size, :args.size], patch)      # warm-up: one window
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, nsum, _ = predict_sliding_window_return_logits(net, vol, patch)
    seg, _ = ops.argmax_dice(acc.permute(3, 0, 1, 2)[None])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nwin = (max(1, -(-(n - args.size) // (args.size // 2))) + 1) ** 3 if n > args.size else 1
    return {"volume": n, "windows": nwin, "ms_per_window": round(dt / nwin * 1e3, 3), "seconds": round(dt, 3),
            "accumulator_gib": round(acc.numel() * 4 / 2 ** 30, 2), "classes": int(acc.shape[-1]),
            "note": "one ensemble member; network forward (4 windows per pass) + Gaussian accumulate + final argmax"}


def product_switches():
    """The environment switches of the product path in force for this run (Switches)."""
    from dg_tta_amd.tta.tta import batch_branches_enabled, batched_steps
    return {"DGTTA_BATCH_BRANCHES": int(batch_branches_enabled()), "steps_per_pass": batched_steps(16, 1),
            "exact_zero_bias_grad": True, "accumulate_grads_in_place": True,
            "env": {k: v for k, v in os.environ.items() if k.startswith("DGTTA_")}}


def roofline_of(probe, args, dtype):
    """Roofline of the dominant kernel from the events recorded around its launches inside the timed region."""
    if not probe["events"]:
        return None
    # launches of the probed block: training passes carry 2 branches x k accumulation steps, the eval pass 1 sample
    times = [(s.elapsed_time(e), nb_) for s, e, nb_ in probe["events"]]
    nb = max(n for _, n in times)
    times = [t for t, n in times if n == nb]
    avg_ms = sum(times) / len(times)
    flop = conv_flops(probe["cin"], probe["cout"], probe["vout"]) * nb
    peak = 157.3 if dtype == "fp32" else 2500.0
    ach = flop / (avg_ms * 1e-3) / 1e12
    traffic, src = None, None       # HBM bytes per launch: rocprofv3 PMC passes committed under profiles/ (not a live counter)
    for cand in ("r02_pmc_summary.json", "r01_pmc_summary.json"):
        pmc = ROOT / "profiles" / cand
        if pmc.exists() and dtype != "fp32" and args.size == 128:
            d = json.loads(pmc.read_text()).get("conv_128cube_32to32", {})
            if "fetch_bytes_corrected_median" in d:     # PMC pass = one sample of this layer; scaled by the batch
                traffic = (d["fetch_bytes_corrected_median"] + d["write_bytes_median"]) * nb
                src = f"profiles/{cand} (rocprofv3 --pmc passes of one sample of this layer x samples_per_launch)"
                break
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_source": src,
            "kernel": "conv3_mfma_kernel" if dtype == "fp32" else "conv3_rows_kernel",
            "launches": len(times), "avg_ms": round(avg_ms, 4), "flop_per_launch": flop, "samples_per_launch": nb,
            "scope": "forward launches of block dec.3.1 (128^3 32->32, fused statistics) in the training passes "
                     "(samples_per_launch = 2 branches x k accumulation steps); the kernel name also runs the other "
                     "large layers, so rocprofv3's per-name average is a mix of shapes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--accum", type=int, default=16)
    ap.add_argument("--copt", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "fp16"],
                    help="activation storage; accumulation, statistics, loss, gradients of weights and AdamW are fp32")
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--cpu-size", type=int, default=128)
    ap.add_argument("--cpu-warmup", type=int, default=1)
    ap.add_argument("--no-fp32", action="store_true", help="skip the nested reference-precision (fp32) epoch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inference-size", type=int, default=256,
                    help="edge of the volume for the sliding-window inference leg (BASELINE config 3: 512); 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = torch.device(f"cuda:{local}")

    from dg_tta_amd.sharding import max_over_ranks
    from dg_tta_amd.unet import set_probe

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(dtype, steps, warmup):
        """W untimed + K timed epochs of the product path in `dtype`; returns (seconds max over ranks, runner, roofline)."""
        args.dtype = dtype
        torch.manual_seed(1234 + rank)
        np.random.seed(1234 + rank)
        runner = EpochRunner(args, device, rank)
        for _ in range(warmup):
            runner.epoch()
        probe = set_probe(("dec", 3, 1))          # the 128^3 32->32 conv block (largest single-shape FLOP share)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.epoch()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0, device)
        set_probe(None)
        return dt, runner, roofline_of(probe, args, dtype)

    main_dtype = args.dtype
    dt, runner, roof = timed_run(main_dtype, args.steps, args.warmup)
    losses, dice = list(runner.losses), runner.dice
    del runner
    torch.cuda.empty_cache()
    other = None
    if not args.no_fp32 and main_dtype != "fp32":
        # the reference never autocasts during TTA: the same epoch with fp32 storage / fp32 MFMA
        fdt, frunner, froof = timed_run("fp32", 1, 1)
        other = {"value": round(world / fdt, 5), "value_per_gpu": round(1.0 / fdt, 5), "unit": "TTA-epochs/s", "steps": 1,
                 "warmup": 1, "ms_per_step": round(fdt * 1e3, 2), "loss_last_epoch": frunner.losses[-1],
                 "pseudo_dice": frunner.dice, "roofline": froof}
        del frunner
        torch.cuda.empty_cache()
    args.dtype = main_dtype

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.steps / dt
        out = {"metric": "TTA-epochs/sec per GPU on 128^3 patch", "value": round(value, 5), "unit": "TTA-epochs/s",
               "value_per_gpu": round(args.steps / dt, 5),
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic",
               "config": {"workload": f"tta_epoch: {args.size}^3 patch from a {args.size + 32}^3 volume, "
                                      f"{args.accum} accumulation steps, GIN+affine in both branches, MIND 12ch, "
                                      f"nnUNet 3d_fullres 105 classes, C_opt={args.copt}, AdamW, 1 eval patch",
                          "patch": args.size, "accum": args.accum, "c_opt": args.copt,
                          "parallelism": f"{world} independent TTA instance(s), sample-sharded",
                          "value_is": "whole-job aggregate over all GPUs (value_per_gpu = one instance)",
                          "product_switches": product_switches()},
               "loss_last_epoch": losses[-1], "pseudo_dice": dice,
               "epoch_tflop": round(96.89 * (args.size / 128) ** 3 * (args.accum * 6 + 1) / 97.0, 2),
               "roofline": roof}
        if other is not None:
            out["fp32"] = other
        if args.inference_size > 0 and world == 1:
            out["inference"] = inference_leg(args, device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
