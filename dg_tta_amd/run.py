"""`dgtta` command line — same sub-commands, positional arguments and option names as the reference's dg_tta/run.py
(prepare_tta :71-118, run_tta :120-209).  `inject_trainers` / `pretrain` dispatch into nnU-Net training and are out of
scope for this engine.  Extra, optional: `run_tta --gpus N` fans out one process per GPU
(sample-sharded, no collectives), `--dtype {fp32,bf16,fp16}` (default fp32 = the reference's precision).
"""
import argparse
import json
import os
import re
import subprocess
import sys
from datetime import datetime
from pathlib import Path

import torch

from .utils import check_dga_root_is_set
from .tta.torch_utils import generate_label_mapping
from .tta.config_log_utils import (check_dataset_pretrain_config, get_tta_folders, load_current_modifier_functions,
                                   prepare_tta as _prepare_tta)

DEFAULT_DTYPE = "fp32"      # activation storage of `run_tta`: the reference's precision (dg_tta/tta/tta.py:560 never autocasts); it is
                            # the setting that keeps north_star's bit-exact label maps (tests/test_gpu_tta.py::test_tta_unit_golden)
FAST_DTYPE = "fp16"         # opt-in 16-bit storage (`--dtype fp16|bf16`).  fp16 (guarded loss scale; BASELINE config 5's mixed precision) is
                            # the headline dtype of bench.py since round 6: the 16-bit type that meets the stated parity tolerances
                            # against the oracle (tests/test_gpu_referee.py) at the same rate as bf16 (BASELINE config 2's name)

_ADJ = ("brisk", "calm", "eager", "fuzzy", "keen", "lucid", "mellow", "nimble", "quiet", "rapid", "solid", "vivid")
_NOUN = ("atlas", "beacon", "cortex", "delta", "ember", "fjord", "gamma", "harbor", "isthmus", "kernel", "lattice", "voxel")


def _random_name():
    """`randomname.get_name()` stand-in (package not installable offline): adjective-noun."""
    import random
    r = random.SystemRandom()
    return f"{r.choice(_ADJ)}-{r.choice(_NOUN)}"


def _add_common(parser):
    parser.add_argument("pretrained_dataset_id", help="Task ID for pretrained model. Can be numeric or one of "
                        "['TS104_GIN', 'TS104_MIND', 'TS104_GIN_MIND']")
    parser.add_argument("tta_dataset_id", help="Task ID for TTA")
    parser.add_argument("--pretrainer", help="Trainer to use for pretraining", default=None)
    parser.add_argument("--pretrainer_config", help="Fold ID of nnUNet model to use for pretraining", default="3d_fullres")
    parser.add_argument("--pretrainer_fold", help="Fold ID of nnUNet model to use for pretraining", default="0")


def _wait_children(procs, poll_s=0.5):
    """Waits for the per-GPU children; when one fails the others are stopped (they would otherwise sit in the filesystem
    barrier waiting for the dead rank's files).  Returns the list of return codes."""
    import time
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            return rcs
        if any(rc is not None and rc != 0 for rc in rcs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            return [p.wait() for p in procs]
        time.sleep(poll_s)


class DGTTAProgram:
    def __init__(self, argv=None):
        self.argv = list(sys.argv if argv is None else argv)
        parser = argparse.ArgumentParser(description="DG-TTA for nnUNetv2 (MI355X engine)", usage="""dgtta <command> [<args>]

        Commands are:
        prepare_tta     Prepare test-time adaptation
        run_tta         Run test-time adaptation
        """)
        parser.add_argument("command", help="Subcommand to run")
        args = parser.parse_args(self.argv[1:2])
        if args.command.startswith("_") or not hasattr(self, args.command):
            print("Unrecognized command")
            parser.print_help()
            raise SystemExit(1)
        getattr(self, args.command)()

    def inject_trainers(self):
        raise SystemExit("inject_trainers patches an installed nnunetv2 for source-domain pre-training; "
                         "use the reference package for that step (out of scope of the TTA engine).")

    def pretrain(self):
        raise SystemExit("pretrain dispatches into nnUNetv2_train; use the reference package (out of scope here).")

    def prepare_tta(self):
        parser = argparse.ArgumentParser(description="Prepare DG-TTA", usage="dgtta prepare_tta [-h]")
        _add_common(parser)
        parser.add_argument("--tta_dataset_bucket", help="Can be one of ['imagesTr', 'imagesTs', 'imagesTrAndTs']",
                            default="imagesTs")
        args = parser.parse_args(self.argv[2:])
        ds, trainer, cfg, fold = check_dataset_pretrain_config(args.pretrained_dataset_id, args.pretrainer,
                                                               args.pretrainer_config, args.pretrainer_fold)
        _prepare_tta(ds, int(args.tta_dataset_id), pretrainer=trainer, pretrainer_config=cfg, pretrainer_fold=fold,
                     tta_dataset_bucket=args.tta_dataset_bucket)

    def run_tta(self):
        parser = argparse.ArgumentParser(description="Run DG-TTA")
        _add_common(parser)
        parser.add_argument("--device", help="Device to be used", default="cuda")
        parser.add_argument("--gpus", type=int, default=1, help="one TTA process per GPU, samples sharded round-robin")
        parser.add_argument("--dtype", choices=["fp32", "bf16", "fp16"], default=DEFAULT_DTYPE,
                            help="activation storage: fp32 (default) = the reference's precision, reproduces its label maps; "
                                 "fp16 / bf16 = opt-in 16-bit storage with fp32 accumulation at ~5.5x the fp32 rate.  Measured on a "
                                 "pre-trained synthetic model against the CPU restatement of the reference's loop "
                                 "(tests/test_gpu_referee.py, profiles/r0*_dice_delta_12_epochs*.json): fp16 (guarded loss scale) "
                                 "stays within 1e-3 of the reference's Dice AND of its per-epoch loss; bf16 holds the Dice on a "
                                 "well-trained model (2e-3 off on a weaker one) but not the loss - prefer fp16; "
                                 "not measured on real TS104 weights (no network here): check on your data")
        parser.add_argument("--run_name", default=None,
                            help="name of the run directory (default: timestamp + random name).  Required, and the same on "
                                 "every rank, when RANK / WORLD_SIZE are set by an external launcher")
        args = parser.parse_args(self.argv[2:])
        ds, trainer, cfg, fold = check_dataset_pretrain_config(args.pretrained_dataset_id, args.pretrainer,
                                                               args.pretrainer_config, args.pretrainer_fold)
        tta_data_dir, plan_dir, results_dir, pre_name, tta_name = get_tta_folders(ds, int(args.tta_dataset_id), trainer,
                                                                                  cfg, fold)
        run_name = args.run_name
        if run_name is None and int(os.environ.get("WORLD_SIZE", 1)) > 1:
            # every rank would invent its own timestamp + random name and then wait for the others in a directory of its own
            raise SystemExit("run_tta: WORLD_SIZE > 1 needs --run_name (the same on every rank); `--gpus N` sets it itself")
        if int(os.environ.get("WORLD_SIZE", 1)) > 1:
            from .sharding import launch_id
            if not launch_id():
                # the done / failed markers of two launches into one run directory could not be told apart
                raise SystemExit("run_tta: WORLD_SIZE > 1 under a launcher other than torch.distributed.run needs DGTTA_LAUNCH_ID "
                                 "(any string that is the same on every rank and new for every launch)")
        if run_name is None:
            now_str = datetime.now().strftime("%Y%m%d__%H_%M_%S")
            results_dir.mkdir(exist_ok=True, parents=True)
            numbers = [int(m[0]) for m in (re.search(r"[0-9]+$", str(p)) for p in results_dir.iterdir()) if m]
            run_no = 0 if len(numbers) == 0 else max(numbers) + 1
            run_name = f"{now_str}_{_random_name()}-{run_no}"

        rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
        if args.gpus > 1 and world == 1:
            # fan out: one independent process per GPU, pinned with HIP_VISIBLE_DEVICES, same run directory
            (results_dir / run_name).mkdir(exist_ok=True, parents=True)
            from .sharding import child_devices
            import uuid
            procs, launch = [], uuid.uuid4().hex
            for r, dev_id in enumerate(child_devices(args.gpus)):
                env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.gpus), HIP_VISIBLE_DEVICES=dev_id,
                           DGTTA_SUMMARY_BY_PARENT="1", DGTTA_LAUNCH_ID=launch)
                env.pop("CUDA_VISIBLE_DEVICES", None)
                cmd = [sys.executable, "-m", "dg_tta_amd.run"] + self.argv[1:] + ["--run_name", run_name]
                procs.append(subprocess.Popen(cmd, env=env))
            rcs = _wait_children(procs)
            if any(rc != 0 for rc in rcs):       # a child killed by a signal has a NEGATIVE return code
                print(f"run_tta: child return codes {rcs}: not evaluating an incomplete run", file=sys.stderr)
                raise SystemExit(1)
            # every child has exited: all predictions are on disk, the summary cannot race (reference: tta.py:447-470)
            with open(Path(plan_dir) / "tta_plan.json", "r") as f:
                config = json.load(f)
            from .tta.tta import evaluate_run
            evaluate_run(results_dir / run_name, config, load_current_modifier_functions(plan_dir), torch.device(args.device))
            raise SystemExit(0)

        with open(Path(plan_dir) / "tta_plan.json", "r") as f:
            config = json.load(f)
        with open(Path(plan_dir) / f"{pre_name}_label_mapping.json", "r") as f:
            pretrained_label_mapping = json.load(f)
        with open(Path(plan_dir) / f"{tta_name}_label_mapping.json", "r") as f:
            tta_dataset_label_mapping = json.load(f)
        label_mapping = generate_label_mapping(pretrained_label_mapping, tta_dataset_label_mapping)
        modifier_fn_module = load_current_modifier_functions(plan_dir)
        from .tta.tta import tta_main
        (results_dir / run_name).parent.mkdir(exist_ok=True, parents=True)
        tta_main(run_name=run_name, config=config, tta_data_dir=tta_data_dir, save_base_path=results_dir,
                 label_mapping=label_mapping, modifier_fn_module=modifier_fn_module, device=torch.device(args.device),
                 shard=(rank, world),
                 act_dtype={"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype])


def main():
    if len(sys.argv) == 1 or sys.argv[1] in ["--help", "-h"]:
        check_dga_root_is_set(soft_check=True)
    else:
        check_dga_root_is_set()
    DGTTAProgram()


if __name__ == "__main__":
    main()
