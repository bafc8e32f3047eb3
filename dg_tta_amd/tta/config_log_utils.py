"""Plan / folder / logging surface of `dgtta prepare_tta|run_tta` — same file names, JSON keys and Python plugin API as
the reference's dg_tta/tta/config_log_utils.py (TEMPLATE_PLAN :24-41, ModifierFunctions :44-70, get_tta_folders
:87-121, check_dataset_pretrain_config :124-187, prepare_tta :190-300, get_global_idx :353-362,
load_current_modifier_functions :365-374, get_data_filepaths :377-394, get_parameters_save_path :463-468).
Host-side only (no kernels).  nnunetv2 is not required: its two helpers used here (dataset-name lookup and the
nnUNet_raw / nnUNet_results paths) are read from the same environment variables nnU-Net uses.
"""
import importlib
import importlib.util
import inspect
import json
import os
import pathlib
import shutil
import sys
from contextlib import contextmanager
from pathlib import Path

import torch

TEMPLATE_PLAN = dict(
    tta_across_all_samples=False,
    tta_eval_patches=1,
    batch_size=1,
    patches_to_be_accumulated=16,
    lr=1e-5,
    ensemble_count=3,
    epochs=12,
    start_tta_at_epoch=1,
    intensity_aug_function="GIN",  # ['GIN', 'disabled']
    spatial_aug_type="affine",  # ['affine', 'deformable']
    params_with_grad="all",  # all, norms, encoder
    have_grad_in="branch_a",  # ['branch_a', 'branch_b', 'both']
    do_intensity_aug_in="none",  # ['branch_a', 'branch_b', 'both', 'none']
    do_spatial_aug_in="both",  # ['branch_a', 'branch_b', 'both', 'none']
    num_processes=1,
    wandb_mode="disabled",
)

TS104_IDS = ("TS104_GIN", "TS104_MIND", "TS104_GIN_MIND", "TS104_GIN_MultiRes", "TS104_MIND_MultiRes",
             "TS104_GIN_MIND_MultiRes")


class ModifierFunctions:
    def __init__(self):
        pass

    @staticmethod
    def modify_tta_input_fn(image: torch.Tensor):
        assert image.ndim == 5  # B,1,D,H,W
        # This function will be called on the input that is fed to the model
        return image

    @staticmethod
    def modfify_tta_model_output_fn(pred_label: torch.Tensor):
        assert pred_label.ndim == 5  # B,C,D,H,W
        # This function will be called directly after model prediction
        return pred_label

    @staticmethod
    def modify_tta_output_after_mapping_fn(mapped_label: torch.Tensor):
        assert mapped_label.ndim == 5  # B,MAPPED_C,D,H,W
        # This function will be called after model prediction when labels are mapped
        # to the target label numbers/ids.
        return mapped_label

    @staticmethod
    def postprocess_results_fn(results_dir: pathlib.Path):
        pass
        # This function will be called on the final output directory.


def is_template_modifier(fn, name):
    """True when a user's modifier function is byte-code identical to the template's (lets the engine fuse
    map_label into the segmentation head, which is only valid when the model-output modifier is the identity)."""
    ref = getattr(ModifierFunctions, name)
    try:
        a, b = fn.__code__, ref.__code__
        return a.co_code == b.co_code and a.co_consts == b.co_consts and a.co_names == b.co_names
    except AttributeError:
        return False


def nnunet_path(var):
    val = os.environ.get(var)
    if val is None:
        raise EnvironmentError(f"{var} is not set (nnU-Net environment variable)")
    return val


def maybe_convert_to_dataset_name(dataset_id):
    """nnU-Net convention: the folder `Dataset{id:03d}_<name>` in nnUNet_raw / _preprocessed / _results."""
    if isinstance(dataset_id, str) and not dataset_id.isnumeric():
        return dataset_id
    prefix = "Dataset%03.0d_" % int(dataset_id)
    found = set()
    for var in ("nnUNet_raw", "nnUNet_preprocessed", "nnUNet_results"):
        root = os.environ.get(var)
        if root and Path(root).is_dir():
            found.update(p.name for p in Path(root).iterdir() if p.is_dir() and p.name.startswith(prefix))
    if len(found) != 1:
        raise RuntimeError(f"dataset id {dataset_id}: expected exactly one folder {prefix}* in the nnU-Net dirs, "
                           f"found {sorted(found)}")
    return found.pop()


def get_tta_folders(pretrained_dataset_id, tta_dataset_id, pretrainer, pretrainer_config, pretrainer_fold):
    root_dir = Path(os.environ["DG_TTA_ROOT"])
    tta_dataset_name = maybe_convert_to_dataset_name(tta_dataset_id)
    if isinstance(pretrained_dataset_id, int):
        pretrained_dataset_name = maybe_convert_to_dataset_name(pretrained_dataset_id)
    else:
        pretrained_dataset_name = pretrained_dataset_id
    fold_folder = f"fold_{pretrainer_fold}" if pretrainer_fold != "all" else pretrainer_fold
    map_folder = f"Pretrained_{pretrained_dataset_name}_at_{tta_dataset_name}"
    pretrainer_folder = f"{pretrainer}__{pretrainer_config}"
    plan_dir = root_dir / "plans" / map_folder / pretrainer_folder / fold_folder
    results_dir = root_dir / "results" / map_folder / pretrainer_folder / fold_folder
    tta_data_dir = Path(nnunet_path("nnUNet_raw"), tta_dataset_name)
    return tta_data_dir, plan_dir, results_dir, pretrained_dataset_name, tta_dataset_name


def check_dataset_pretrain_config(pretrained_dataset_id, pretrainer, pretrainer_config, pretrainer_fold):
    if pretrained_dataset_id.isnumeric():
        pretrained_dataset_id = int(pretrained_dataset_id)
    if str(pretrainer_fold).isnumeric():
        pretrainer_fold = int(pretrainer_fold)
    assert pretrained_dataset_id in TS104_IDS or isinstance(pretrained_dataset_id, int)
    if isinstance(pretrained_dataset_id, int):
        assert pretrainer is not None
        assert pretrainer_config is not None
        assert pretrainer_fold == "all" or isinstance(pretrainer_fold, int)
    else:
        # the six downloadable TotalSegmentator-104 models: trainer name is the id with the TS104 prefix swapped
        pretrainer = "nnUNetTrainer_" + pretrained_dataset_id[len("TS104_"):]
        pretrainer_config = "3d_fullres"
        pretrainer_fold = "0"
    return pretrained_dataset_id, pretrainer, pretrainer_config, pretrainer_fold


def get_resources_dir():
    return Path(__file__).resolve().parents[1] / "__resources__"


def download_pretrained_weights(pretrained_dataset_id):
    """Lays out `$DG_TTA_ROOT/_pretrained_weights/<trainer>__nnUNetPlans__3d_fullres/{plans,dataset}.json, fold_0/`.
    The checkpoint itself must be placed at fold_0/checkpoint_final.pth by the user when offline (the reference
    fetches it with wget from cloud.imi.uni-luebeck.de, config_log_utils.py:307-350)."""
    if pretrained_dataset_id not in TS104_IDS:
        raise ValueError(pretrained_dataset_id)
    pretrainer_dir = f"nnUNetTrainer_{pretrained_dataset_id[len('TS104_'):]}__nnUNetPlans__3d_fullres"
    target = Path(os.environ["DG_TTA_ROOT"]) / "_pretrained_weights" / pretrainer_dir
    weights = target / "fold_0" / "checkpoint_final.pth"
    weights.parent.mkdir(exist_ok=True, parents=True)
    skeleton = get_resources_dir() / "model_skeleton"
    for name in ("plans.json", "dataset.json"):
        if not (target / name).exists() and (skeleton / name).exists():
            shutil.copy(skeleton / name, target / name)
    if not weights.exists():
        print(f"Pretrained weights not found at {weights}: copy the TS104 checkpoint there (no network access here).")
    return target, weights


def prepare_tta(pretrained_dataset_id, tta_dataset_id, pretrainer, pretrainer_config, pretrainer_fold,
                tta_dataset_bucket="imagesTs"):
    root_dir = Path(os.environ["DG_TTA_ROOT"])
    assert root_dir.is_dir()
    _, plan_dir, results_dir, pretrained_dataset_name, tta_dataset_name = get_tta_folders(
        pretrained_dataset_id, tta_dataset_id, pretrainer, pretrainer_config, pretrainer_fold)
    shutil.rmtree(plan_dir, ignore_errors=True)
    plan_dir.mkdir(exist_ok=True, parents=True)
    results_dir.mkdir(exist_ok=True, parents=True)

    if isinstance(pretrained_dataset_id, str):
        target_path, weights_file_path = download_pretrained_weights(pretrained_dataset_id)
        with open(target_path / "dataset.json", "r") as f:
            pretrained_classes = json.load(f)["labels"]
    else:
        with open(Path(nnunet_path("nnUNet_raw"), pretrained_dataset_name) / "dataset.json", "r") as f:
            pretrained_classes = json.load(f)["labels"]
        fold_dir = f"fold_{pretrainer_fold}" if pretrainer_fold != "all" else pretrainer_fold
        weights_file_path = Path(nnunet_path("nnUNet_results"), pretrained_dataset_name,
                                 f"{pretrainer}__nnUNetPlans__{pretrainer_config}", fold_dir, "checkpoint_final.pth")
        if not weights_file_path.is_file():
            raise FileNotFoundError(f"Could not find weights file at {weights_file_path}")

    with open(Path(nnunet_path("nnUNet_raw"), tta_dataset_name) / "dataset.json", "r") as f:
        tta_dataset_classes = json.load(f)["labels"]

    with open(plan_dir / f"{pretrained_dataset_name}_label_mapping.json", "w") as f:
        json.dump(pretrained_classes, f, indent=4)
    with open(plan_dir / f"{tta_dataset_name}_label_mapping.json", "w") as f:
        json.dump(tta_dataset_classes, f, indent=4)

    plan = TEMPLATE_PLAN.copy()
    plan["__pretrained_dataset_name__"] = pretrained_dataset_name
    plan["__tta_dataset_name__"] = tta_dataset_name
    plan["pretrained_weights_filepath"] = str(weights_file_path)
    common = sorted(set(pretrained_classes.keys()).intersection(set(tta_dataset_classes)))
    assert "background" in common, "Background class must be present in both datasets!"
    common.remove("background")
    plan["optimized_labels"] = ["background"] + common
    plan["tta_data_filepaths"] = [str(fp) for fp in get_data_filepaths(tta_dataset_name, tta_dataset_bucket)]
    with open(plan_dir / "tta_plan.json", "w") as f:
        json.dump(plan, f, indent=4)

    with open(plan_dir / "modifier_functions.py", "w") as f:
        f.write("import pathlib\n")
        f.write("import torch\n\n")
        f.write(inspect.getsource(ModifierFunctions))
    nb = get_resources_dir() / "check_tta_input.ipynb"
    if nb.exists():
        shutil.copyfile(nb, plan_dir / nb.name)
    print(f"\nPreparation done. You can edit the plan, modifier functions and optimized labels in {plan_dir} "
          f"prior to running TTA.")


def get_global_idx(list_of_tuple_idx_max):
    """Decimal-packed step id, smallest identifier last: [(2,3),(250,1000)] -> 2250."""
    global_idx, mult = 0, 1
    for idx, idx_max in reversed(list_of_tuple_idx_max):
        global_idx += mult * idx
        mult *= 10 ** len(str(int(idx_max)))
    return global_idx


def load_current_modifier_functions(plan_dir):
    mod_path = Path(plan_dir) / "modifier_functions.py"
    spec = importlib.util.spec_from_file_location("dg_tta.current_modifier_functions", mod_path)
    dyn_mod = importlib.util.module_from_spec(spec)
    sys.modules["dg_tta.current_modifier_functions"] = dyn_mod
    spec.loader.exec_module(dyn_mod)
    return dyn_mod


def get_data_filepaths(tta_dataset_name, tta_dataset_bucket):
    raw = Path(nnunet_path("nnUNet_raw"), tta_dataset_name)
    buckets = {"imagesTr": ["imagesTr"], "imagesTs": ["imagesTs"], "imagesTrAndTs": ["imagesTr", "imagesTs"]}
    files = []
    for sub in buckets[tta_dataset_bucket]:
        if (raw / sub).is_dir():
            files.extend(sorted(p for p in (raw / sub).iterdir() if p.is_file()))
    return files


def wandb_run_is_available():
    if importlib.util.find_spec("wandb") is None:
        return False
    import wandb
    return wandb.run is not None and not wandb.run.disabled


@contextmanager
def suppress_stdout():
    with open(os.devnull, "w") as devnull:
        old, sys.stdout = sys.stdout, devnull
        try:
            yield
        finally:
            sys.stdout = old


def plot_run_results(save_path, sample_id, ensemble_idx, tta_losses, eval_dices):
    """Loss / pseudo-Dice curve PNG `<case>__ensemble_idx_<k>_tta_results.png` (config_log_utils.py:426-452)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
    except Exception:      # plotting is optional
        return
    fig, ax1 = plt.subplots()
    ax2 = ax1.twinx()
    ax1.plot(tta_losses, label="loss", c="#e7475e")
    ax1.set_xlim(0, max(len(tta_losses) - 1, 1))
    ax1.set_ylabel("Soft-Dice Loss")
    ax1.set_xlabel("TTA Epoch")
    ax2.plot(eval_dices * 100, label="eval_dices", c="#248888")
    ax2.set_ylabel("Pseudo-Dice in %")
    fig.suptitle(f"{sample_id} (ensemble_idx={ensemble_idx})")
    name = sample_id.split("/")[-1]
    fig.savefig(Path(save_path) / f"{name}__ensemble_idx_{ensemble_idx}_tta_results.png")
    plt.close(fig)


def get_parameters_save_path(save_path, sample_id, ensemble_idx):
    sample_id = sample_id.split("/")[-1]
    return Path(save_path) / f"{sample_id}__ensemble_idx_{ensemble_idx}_tta_parameters.pt"
