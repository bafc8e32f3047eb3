"""Bridge to the nnU-Net model folder and target data WITHOUT nnunetv2 (not installable offline) — the role of the
reference's dg_tta/tta/nnunet_utils.py (load_network :88-113, load_tta_data :63-85, preprocess_fromfile :170-204).

* `load_network` reads `<model_folder>/{plans.json,dataset.json}` + `fold_k/checkpoint_final.pth` (nnU-Net layout),
  builds a HipPlainConvUNet with the plans' topology and registers the trainer's forward pre-hooks in the reference's
  order (gin_hook, then mind_hook: dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:55-57).
* `load_tta_data` yields `{"data": FloatTensor[1+K,D,H,W], "data_properties", "ofile"}` items like
  preprocess_fromfile.  Raw NIfTI cases are cropped, normalised and resampled as nnU-Net's DefaultPreprocessor does
  (tta/preprocessing.py); `.npy`, `.npz` (key `data`) and `.pt` files are taken as already preprocessed arrays.
"""
import json
import re
from itertools import chain
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from ..gin import gin_hook
from ..mind import mind_hook
from ..unet import HipPlainConvUNet
from ..utils import enable_internal_augmentation


def unet_cfg_from_plans(plans, dataset_json, configuration, in_channels):
    conf = plans["configurations"][configuration]
    while "inherits_from" in conf:
        parent = dict(plans["configurations"][conf["inherits_from"]])
        parent.update({k: v for k, v in conf.items() if k != "inherits_from"})
        conf = parent
    if conf.get("UNet_class_name", "PlainConvUNet") != "PlainConvUNet":
        raise NotImplementedError(f"only PlainConvUNet is built, plans ask for {conf['UNet_class_name']}")
    pools = conf["pool_op_kernel_sizes"]
    if any(len(set(p)) != 1 or p[0] not in (1, 2) for p in pools) or any(k != [3, 3, 3] for k in conf["conv_kernel_sizes"]):
        raise NotImplementedError("only isotropic stride 1/2 and 3x3x3 kernels are built")
    base, cap = conf["UNet_base_num_features"], conf["unet_max_num_features"]
    feats = tuple(min(base * 2 ** i, cap) for i in range(len(pools)))
    return dict(features=feats, strides=tuple(p[0] for p in pools), n_conv_enc=tuple(conf["n_conv_per_stage_encoder"]),
                n_conv_dec=tuple(conf["n_conv_per_stage_decoder"]), in_channels=in_channels,
                num_classes=len(dataset_json["labels"])), list(conf["patch_size"])


def trainer_hooks(trainer_name):
    """(input channels, [pre-hooks]) for the DG trainers (nnUNetTrainer_{GIN,MIND,GIN_MIND}[_MultiRes].py:38-59)."""
    name = trainer_name.replace("_MultiRes", "")
    if name.endswith("GIN_MIND"):
        return 12, [gin_hook, mind_hook]
    if name.endswith("MIND"):
        return 12, [mind_hook]
    if name.endswith("GIN"):
        return 1, [gin_hook]
    raise NotImplementedError(f"trainer {trainer_name}: only the DG-TTA trainers (GIN / MIND / GIN_MIND) are built")


def _load_checked(network, state, weights_file, trainer_name, configuration):
    """network.load_state_dict with a message that says WHAT differs: the key layout expected here is that of
    dynamic-network-architectures' PlainConvUNet (encoder.stages.S.0.convs.K.{conv,norm,all_modules.{0,1}}.*, decoder.encoder.*
    as a second name of the encoder, decoder.stages / transpconvs / seg_layers), written from memory of version 0.2 - a real
    checkpoint_final.pth has not been reachable offline, so a mismatch must be readable at a glance."""
    expected = network.state_dict()
    missing = sorted(k for k in expected if k not in state)
    unexpected = sorted(k for k in state if k not in expected)
    shapes = sorted(f"{k}: file {tuple(state[k].shape)} vs network {tuple(expected[k].shape)}" for k in expected
                    if k in state and hasattr(state[k], "shape") and tuple(state[k].shape) != tuple(expected[k].shape))
    if missing or unexpected or shapes:
        def head(v):
            return ", ".join(v[:6]) + (f", ... ({len(v)} in all)" if len(v) > 6 else "")
        raise RuntimeError(
            f"{weights_file}: the checkpoint does not fit the network built from plans.json ({trainer_name}, {configuration}: "
            f"{len(expected)} tensors expected, {len(state)} in the file).\n"
            f"  missing in the file ({len(missing)}): {head(missing) or '-'}\n"
            f"  not known to the network ({len(unexpected)}): {head(unexpected) or '-'}\n"
            f"  shape differs ({len(shapes)}): {head(shapes) or '-'}\n"
            f"  expected layout: dynamic-network-architectures PlainConvUNet state dict (encoder.stages.<s>.0.convs.<k>.conv.weight, "
            f"...norm.weight, ...all_modules.<0|1>.*, decoder.encoder.* duplicating the encoder, decoder.stages.*, decoder.transpconvs.*, "
            f"decoder.seg_layers.*); a '_orig_mod.' prefix (torch.compile) is stripped")
    network.load_state_dict(state)


def load_network(weights_file, device, act_dtype=torch.float32, conv_impl=0):
    weights_file = Path(weights_file)
    model_folder = weights_file.parents[1]
    configuration = model_folder.name.split("__")[-1]
    trainer_name = model_folder.name.split("__")[0]
    with open(model_folder / "plans.json") as f:
        plans = json.load(f)
    with open(model_folder / "dataset.json") as f:
        dataset_json = json.load(f)
    checkpoint = torch.load(weights_file, map_location="cpu", weights_only=False)
    if isinstance(checkpoint, dict) and "network_weights" in checkpoint:
        trainer_name = checkpoint.get("trainer_name", trainer_name)
        state = checkpoint["network_weights"]
    else:
        state = checkpoint
    in_ch, hooks = trainer_hooks(trainer_name)
    cfg, patch_size = unet_cfg_from_plans(plans, dataset_json, configuration, in_ch)
    network = HipPlainConvUNet(cfg, act_dtype=act_dtype, conv_impl=conv_impl)
    state = {k.replace("_orig_mod.", ""): v for k, v in state.items()}
    _load_checked(network, state, weights_file, trainer_name, configuration)
    enable_internal_augmentation()          # as build_network_architecture does; tta_main switches it off again
    for h in hooks:
        network.register_forward_pre_hook(h)
    predictor = SimpleNamespace(plans=plans, dataset_json=dataset_json, configuration=configuration,
                                trainer_name=trainer_name, device=torch.device(device), network=network,
                                list_of_parameters=[state], patch_size=patch_size)
    return predictor, patch_size, network, [state]


_CASE_RE = re.compile(r"(.*)_[0-9]{4}$")


def _read_array(path):
    path = Path(path)
    if path.suffix == ".npy":
        return torch.from_numpy(np.load(path))
    if path.suffix == ".npz":
        return torch.from_numpy(np.load(path)["data"])
    if path.suffix == ".pt":
        return torch.load(path, map_location="cpu")
    raise NotImplementedError(f"{path.name}: reading raw medical image formats needs SimpleITK + nnU-Net preprocessing "
                              f"(SURVEY.md §8f 'next'); provide preprocessed .npy/.npz/.pt arrays [C,D,H,W]")


def _is_nifti(path):
    """A raw volume file this engine reads itself: NIfTI, and (round 5) NRRD / MetaImage (tta/image_io.py)."""
    from .image_io import is_image_file
    return is_image_file(path)


def preprocess_fromfile(image_file, label_file, ofile, predictor=None):
    """dg_tta/tta/nnunet_utils.py:170-204: preprocessed image channel(s) + one one-hot channel per foreground label.
    NIfTI cases go through the restated DefaultPreprocessor (tta/preprocessing.py: crop, normalise, resample to the
    plans' spacing; resampling on the GPU); `.npy/.npz/.pt` cases are taken as already preprocessed arrays."""
    if _is_nifti(image_file):
        if predictor is None or not hasattr(predictor, "plans"):
            raise RuntimeError("raw NIfTI cases need the model's plans.json (load_network's predictor) for preprocessing")
        from .preprocessing import run_case
        lbl = label_file if (label_file is not None and Path(label_file).is_file()) else None
        data, seg, props = run_case([image_file], lbl, predictor.plans, predictor.configuration,
                                    getattr(predictor, "device", "cuda"))
        img = torch.from_numpy(np.ascontiguousarray(data)).float()
        out = img
        if lbl is not None:
            k = int(max(seg.max(), 0))
            seg_t = torch.from_numpy(seg[0].astype(np.int64))
            onehot = torch.stack([(seg_t == i + 1).float() for i in range(k)]) if k > 0 else torch.zeros((0, *seg_t.shape))
            out = torch.cat([img, onehot], 0)
        return {"data": out.contiguous().float(), "data_properties": props, "ofile": ofile}
    img = _read_array(image_file).float()
    if img.dim() == 3:
        img = img[None]
    data = img
    if label_file is not None and Path(label_file).is_file():
        seg = _read_array(label_file)
        seg = seg[0] if seg.dim() == 4 else seg
        k = int(seg.max())
        onehot = torch.stack([(seg == i + 1).float() for i in range(k)]) if k > 0 else torch.zeros((0, *seg.shape))
        data = torch.cat([img[:1], onehot], 0)
    return {"data": data.contiguous().float(), "data_properties": {"shape": tuple(img.shape[1:])}, "ofile": ofile}


def get_data_iterator(tta_data_filepaths, dataset_raw_path, bucket, predictor=None, wanted=None, first_index=0,
                      total=None):
    """Lazy iterator over the bucket's cases.  wanted(global_index, total) -> bool: cases it rejects are NOT read or
    preprocessed; a stub {"ofile", "skipped": True} takes their place so that the consumer's indexing is unchanged
    (multi-GPU runs: every rank walks the same file list and loads only the cases it works on)."""
    assert bucket in ("imagesTs", "imagesTr")
    files = [Path(p) for p in tta_data_filepaths if Path(p).parts[-2] == bucket]
    label_folder = Path(dataset_raw_path) / ("labelsTs" if bucket == "imagesTs" else "labelsTr")
    out_folder = "tta_outputTs" if bucket == "imagesTs" else "tta_outputTr"

    def gen():
        for k, f in enumerate(files):
            stem = f.name[: -len("".join(f.suffixes))] if f.suffixes else f.name
            m = _CASE_RE.match(stem)
            case = m.group(1) if m else stem
            if wanted is not None and not wanted(first_index + k, total if total is not None else len(files)):
                yield {"ofile": f"{out_folder}/{case}", "skipped": True}
                continue
            lbl = label_folder / (case + "".join(f.suffixes))
            yield preprocess_fromfile(f, lbl if lbl.is_file() else None, f"{out_folder}/{case}", predictor)

    return gen(), len(files)


def load_tta_data(config, dataset_raw_path, predictor=None, tta_across_all_samples=False, wanted=None):
    paths = config["tta_data_filepaths"]
    n_ts = sum(1 for p in paths if Path(p).parts[-2] == "imagesTs")
    n_tr = sum(1 for p in paths if Path(p).parts[-2] == "imagesTr")
    ts_it, ts_n = get_data_iterator(paths, dataset_raw_path, "imagesTs", predictor, wanted, 0, n_ts + n_tr)
    tr_it, tr_n = get_data_iterator(paths, dataset_raw_path, "imagesTr", predictor, wanted, n_ts, n_ts + n_tr)
    if tta_across_all_samples:
        return list(ts_it) + list(tr_it), ts_n + tr_n
    return chain(ts_it, tr_it), ts_n + tr_n
