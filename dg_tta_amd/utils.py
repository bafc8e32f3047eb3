"""Environment flags of the reference (dg_tta/utils.py:5-30), same names and semantics."""
import os
import threading
from pathlib import Path

import numpy as np
import torch


def enable_internal_augmentation():
    os.environ["DG_TTA_INTERNAL_AUGMENTATION"] = "true"


def disable_internal_augmentation():
    os.environ["DG_TTA_INTERNAL_AUGMENTATION"] = "false"


def check_internal_augmentation_disabled():
    assert os.environ.get("DG_TTA_INTERNAL_AUGMENTATION", "false").lower() != "true"


def get_internal_augmentation_enabled():
    return os.environ.get("DG_TTA_INTERNAL_AUGMENTATION", "false").lower() == "true"


def check_dga_root_is_set(soft_check=False):
    prompt = "Please define an existing root directory for DG-TTA by setting DG_TTA_ROOT."
    ok = Path(os.environ.get("DG_TTA_ROOT", "_")).is_dir()
    if soft_check and not ok:
        print(prompt)
        return
    assert ok, prompt


def set_environ_vars_from_paths_sh(sh_path):
    with open(sh_path, "r") as f:
        for line in f.readlines():
            if "=" not in line:
                continue
            k, v = line.replace("export", "").split("=", 1)
            os.environ[k.strip()] = v.strip().replace('"', "").replace("'", "")


def upload_async(cpu_tensors, device):
    """Moves a list of small fp32 CPU tensors to `device` with ONE non-blocking copy from pinned staging memory and
    returns device views of the original shapes.  A plain `.to(device)` from pageable memory blocks the host until the
    stream reaches the copy, i.e. every per-branch draw (GIN kernels, affine matrices, patch offsets) used to drain the
    launch queue; the TTA loop makes ~10 of them per branch."""
    device = torch.device(device)
    if device.type != "cuda":
        return [t.to(device) for t in cpu_tensors]
    sizes = [t.numel() for t in cpu_tensors]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4                      # 16-byte aligned pieces
    stage = torch.empty(max(total, 1), dtype=torch.float32, pin_memory=True)
    for t, o, n in zip(cpu_tensors, offs, sizes):
        stage[o:o + n].copy_(t.reshape(-1))
    dev = stage.to(device, non_blocking=True)
    return [dev[o:o + n].view(t.shape) for t, o, n in zip(cpu_tensors, offs, sizes)]


# ---- random draws.  The reference draws from the process-global generators (torch CPU, torch device, numpy legacy), and so
# does this engine by default.  Two TTA instances in ONE process (threads) would interleave their draws on those; rng_scope
# gives the calling thread generators of its own, which every draw site of the path uses (GIN, affine, patch offsets,
# sample choice, MIND noise).
_RNG = threading.local()


class rng_scope:
    """Context: the calling thread's TTA draws come from these generators instead of the global ones.
    cpu: torch.Generator (CPU); device: torch.Generator on the GPU; numpy: np.random.RandomState."""

    def __init__(self, cpu=None, device=None, numpy=None):
        self.new = (cpu, device, numpy)

    def __enter__(self):
        self.prev = getattr(_RNG, "gens", (None, None, None))
        _RNG.gens = self.new
        return self

    def __exit__(self, *exc):
        _RNG.gens = self.prev


def cpu_generator():
    return getattr(_RNG, "gens", (None, None, None))[0]


def device_generator():
    return getattr(_RNG, "gens", (None, None, None))[1]


def numpy_rng():
    r = getattr(_RNG, "gens", (None, None, None))[2]
    return np.random if r is None else r
