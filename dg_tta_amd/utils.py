"""Environment flags of the reference (dg_tta/utils.py:5-30), same names and semantics."""
import os
from pathlib import Path


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
