import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    z = np.load(GOLDEN / f"{name}.npz")
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def unpack_draws(g, prefix):
    """(alpha, ks, kers, shifts), affine, noise  from a golden dict written by make_golden.pack()."""
    ks = [int(k) for k in g[f"{prefix}_ks"]]
    gin = (g[f"{prefix}_alpha"], ks, [g[f"{prefix}_ker{i}"] for i in range(4)],
           [g[f"{prefix}_shift{i}"] for i in range(4)])
    return dict(gin_draw=gin, affine_draw=g[f"{prefix}_affine"], mind_noise=g[f"{prefix}_noise"])


SMALL_CFG = dict(features=(8, 16, 24), strides=(1, 2, 2), n_conv_enc=(2, 2, 2), n_conv_dec=(2, 2),
                 in_channels=12, num_classes=9)


def state_from_golden(g, prefix="w::"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def reload_kernel_switches():
    """The library snapshots the DGTTA_* diagnostic switches once (std::call_once); tests that flip one inside the
    process take a fresh snapshot after changing the environment."""
    from dg_tta_amd import _lib
    _lib.load().dgtta_reload_env()


@pytest.fixture(autouse=True)
def _fresh_kernel_switches():
    yield
    try:                                   # after monkeypatch has restored the environment
        reload_kernel_switches()
    except Exception:
        pass
