"""TTA driver — the per-sample / per-ensemble / per-epoch / accumulation loops of the reference's
dg_tta/tta/tta.py (tta_main :93-373, calc_branch :480-579), with every tensor op executed by the HIP kernels:

    get_batch (csrc/warp.hip) -> 2x calc_branch {GIN (gin.hip) -> affine warp, border (warp.hip) -> MIND pre-hook
    (mind3d.hip) -> PlainConvUNet fwd (unet_ref.hip / conv_mfma.hip) with map_label fused into the head -> inverse
    warp, zeros (warp.hip)} -> masked softmax + soft-Dice (softdice.hip) -> backward through both branches ->
    AdamW once per epoch (adamw.hip).

Behaviour kept bug-compatible with the reference: `have_grad_in` enables/disables autograd for BOTH
branches (tta.py:496-498); epochs before `start_tta_at_epoch` only evaluate the loss; background is excluded from the
loss mean; AdamW receives all parameters.  Multi-GPU: samples are sharded round-robin over ranks (one process per GPU,
no collectives; `shard=(rank, world)`), every (sample, ensemble) unit being independent (tta.py:157-182).
"""
import json
from contextlib import nullcontext
from pathlib import Path

import numpy as np
import torch

from .. import ops
from ..gin import gin_aug
from ..optim import HipAdamW
from ..sharding import (done_marker, failed_marker, mark_rank_done, mark_rank_failed, summary_by_parent, unit_owner,
                        wait_for_done_markers, wait_for_files)
from .._state import state_of
from ..utils import disable_internal_augmentation, numpy_rng, upload_async
from .augmentation_utils import get_rand_affine
from .config_log_utils import (get_global_idx, get_parameters_save_path, is_template_modifier, plot_run_results)
from .model_utils import apply_running_stats, buffer_running_stats, get_model_from_network
from .torch_utils import (dice_coeff, fix_all, get_batch, get_map_idxs, map_label, release_all, release_norms,
                          release_resident)

INTENSITY_AUG_FUNCTION_DICT = {"disabled": lambda img: img, "GIN": gin_aug}
START_CLASS = 1  # Do not use background for consistency loss (tta.py:103)


def repair_ofilename_and_add_fileextension(config, sample):
    """tta.py:57-69: recovers `<case>` and the file extension from the plan's file list (names with `_0000`)."""
    import re
    ofile = sample["ofile"]
    sample.setdefault("file_extension", "")
    for _path in config.get("tta_data_filepaths", []):
        _path = Path(_path)
        prefix, name = "/".join(ofile.split("/")[:-1]), ofile.split("/")[-1]
        if name in str(_path):
            suffixes = "".join(_path.suffixes)
            m = re.match(r"(.*)_[0-9]{4}" + re.escape(suffixes), _path.name)
            if m:
                sample["ofile"] = prefix + "/" + m.group(1)
                sample["file_extension"] = suffixes


def get_sample_specs(config, smp_idx, tta_data, save_path, across_all_samples=False):
    if across_all_samples:
        for e in tta_data:
            repair_ofilename_and_add_fileextension(config, e)
        return None, [e["data"] for e in tta_data], "all_samples", None, save_path / "tta_output"
    sample = next(tta_data)         # a case this rank does not work on arrives as a stub without "data" (not loaded)
    repair_ofilename_and_add_fileextension(config, sample)
    sample_id = sample["ofile"]
    return sample, [sample.get("data")], sample_id, sample["file_extension"], save_path / Path(sample_id).parent


def calc_branch(branch_id, config, model, intensity_aug_func, identity_grid, patch_size, batch_size, label_mapping,
                optimized_labels, modifier_fn_module, imgs, device, head_is_fused=False):
    """One augmented forward pass mapped back to the common frame (reference: tta.py:480-579).
    `identity_grid` is unused (the affine grid is evaluated analytically inside the warp kernel) and kept for API parity."""
    assert branch_id in ["branch_a", "branch_b"]
    grad_context = nullcontext if config["have_grad_in"] in ["branch_a", "both"] else torch.no_grad
    after_mapping = modifier_fn_module.ModifierFunctions.modify_tta_output_after_mapping_fn
    with grad_context():
        imgs_aug = imgs
        if config["do_intensity_aug_in"] in [branch_id, "both"]:
            imgs_aug = intensity_aug_func(imgs_aug)
        spatial = config["do_spatial_aug_in"] in [branch_id, "both"]
        R_inverse = None
        if spatial:
            if config["spatial_aug_type"] == "affine":
                R, R_inverse = get_rand_affine(batch_size, flip=False)
                R, R_inverse = upload_async([R.float().contiguous(), R_inverse.float().contiguous()], device)
            elif config["spatial_aug_type"] == "deformable":
                from .augmentation_utils import get_disp_field
                get_disp_field()
            else:   # no displacement: identity resampling, as the reference's zero grid does
                R = R_inverse = torch.eye(4, device=device)[:3][None].repeat(batch_size, 1, 1)
            imgs_aug = ops.affine_warp(imgs_aug, R, padding_mode="border", tta_grid_algebra=True)
        model.apply(buffer_running_stats if branch_id == "branch_a" else apply_running_stats)
        branch_target = model(imgs_aug)
        if isinstance(branch_target, tuple):
            branch_target = branch_target[0]
        if not head_is_fused:
            branch_target = map_label(branch_target,
                                      get_map_idxs(label_mapping, optimized_labels, input_type="pretrain_labels"),
                                      input_format="logits")
        branch_target = after_mapping(branch_target)
        if isinstance(branch_target, tuple):
            branch_target = branch_target[0]
        if spatial:
            branch_target = ops.affine_warp(branch_target, R_inverse, padding_mode="zeros", tta_grid_algebra=True)
        return branch_target


def calc_both_branches(config, model, intensity_aug_func, patch_size, batch_size, label_mapping, optimized_labels,
                       modifier_fn_module, imgs, device, head_is_fused=False, steps=1):
    """calc_branch("branch_a") and calc_branch("branch_b") with the two network passes run as ONE batch of
    2*batch_size (same weights, per-sample InstanceNorm: the same maths; the weight gradient then sums both branches
    inside one kernel).  Small layers (16^3 and below) are launch / occupancy bound, so halving the number of launches
    and doubling the work per launch is worth ~15 % of an epoch.  Random draws keep the reference's order on both
    generators: branch a's GIN, affine and MIND-noise draws, then branch b's (the noise is pre-drawn and handed to
    mind_hook).  Returns (target_a, target_b).

    steps > 1 additionally batches `steps` accumulation steps (they all see the same weights: the optimizer only steps
    once per epoch): `imgs` is then a callable that samples the next step's patches (get_batch), called in step order so
    that the reference's draw sequence  get_batch_i, branch_a_i, branch_b_i, get_batch_i+1, ...  is kept.  The targets are
    ordered [step][batch item]; ops.consistency_loss on them returns the MEAN of the per-step losses and dice[step]."""
    prepared = prepare_both_branches(config, model, intensity_aug_func, batch_size, imgs, device, steps=steps)
    return run_both_branches(prepared, config, model, label_mapping, optimized_labels, modifier_fn_module, head_is_fused,
                             steps=steps)


def prepare_both_branches(config, model, intensity_aug_func, batch_size, imgs, device, steps=1, precompute_mind=False):
    """The input side of calc_both_branches: patch sampling, intensity / spatial augmentation and the MIND noise draws of
    `steps` accumulation steps x 2 branches, in the reference's draw order.  Holds no reference to the weights, so
    tta_epoch runs it for pass k+1 on a side stream while pass k is still in its backward.  precompute_mind: also
    evaluate the MIND descriptor here (it is handed to the model's mind_hook by run_both_branches)."""
    from ..mind import MIND3D, draw_noise_, uses_mind_hook
    with torch.no_grad():
        augs, inverses, inverses_cpu = [[], []], [[], []], [[], []]
        want_noise = uses_mind_hook(model)
        noise = None
        for step in range(steps):
            step_imgs = imgs() if callable(imgs) else imgs
            for k, branch_id in enumerate(("branch_a", "branch_b")):
                imgs_aug = step_imgs
                if config["do_intensity_aug_in"] in [branch_id, "both"]:
                    imgs_aug = intensity_aug_func(imgs_aug)
                R_inverse = R_inverse_cpu = None
                if config["do_spatial_aug_in"] in [branch_id, "both"]:
                    if config["spatial_aug_type"] == "affine":
                        R, R_inverse = get_rand_affine(batch_size, flip=False)
                        R_inverse_cpu = R_inverse.float().contiguous()
                        R, R_inverse = upload_async([R.float().contiguous(), R_inverse_cpu], device)
                    elif config["spatial_aug_type"] == "deformable":
                        from .augmentation_utils import get_disp_field
                        get_disp_field()
                    else:
                        R = R_inverse = torch.eye(4, device=device)[:3][None].repeat(batch_size, 1, 1)
                    imgs_aug = ops.affine_warp(imgs_aug, R, padding_mode="border", tta_grid_algebra=True)
                augs[k].append(imgs_aug)
                inverses[k].append(R_inverse)
                inverses_cpu[k].append(R_inverse_cpu)
                if want_noise:      # this branch's torch.randn(...) of mind.py:150, drawn in place into its slot of the batch
                    nb_ = imgs_aug.shape[0]
                    if noise is None:
                        noise = torch.empty((2 * steps * nb_, 12) + tuple(imgs_aug.shape[2:]), dtype=torch.float32,
                                            device=imgs_aug.device)
                    slot = k * steps + step
                    draw_noise_(noise[slot * nb_:(slot + 1) * nb_])
        x = torch.cat(augs[0] + augs[1], dim=0)     # all of branch a (step order), then all of branch b
        feat = None
        if want_noise and precompute_mind:
            feat = MIND3D().forward(x, noise=noise, out_dtype=getattr(model, "act_dtype", torch.float32), groups=2 * steps)
            noise = None
    return {"x": x, "inverses": inverses[0] + inverses[1], "inverses_cpu": inverses_cpu[0] + inverses_cpu[1], "noise": noise,
            "feat": feat, "per": augs[0][0].shape[0], "want_noise": want_noise}


def run_both_branches(prepared, config, model, label_mapping, optimized_labels, modifier_fn_module, head_is_fused=False,
                      steps=1):
    """The network side of calc_both_branches on inputs from prepare_both_branches.  Returns (target_a, target_b)."""
    from ..mind import clear_noise, push_features, push_noise
    grad_context = nullcontext if config["have_grad_in"] in ["branch_a", "both"] else torch.no_grad
    after_mapping = modifier_fn_module.ModifierFunctions.modify_tta_output_after_mapping_fn
    inverses, noise, want_noise = prepared["inverses"], prepared["noise"], prepared["want_noise"]
    with grad_context():
        model.apply(buffer_running_stats)
        model.apply(apply_running_stats)
        if prepared["feat"] is not None:
            push_features(model, prepared["feat"])
        elif want_noise:
            push_noise(model, noise, groups=2 * steps)
        per = prepared["per"]
        nb = steps * per
        template_after = is_template_modifier(after_mapping, "modify_tta_output_after_mapping_fn")
        fast = head_is_fused and template_after and all(r is not None for r in inverses)
        # fast path: one inverse warp over the batch, targets are views of one tensor (the loss then runs on it directly and
        # returns one gradient buffer); where the network offers it, that warp is fused into the head (no un-warped logits)
        warp_in_head = nullcontext()
        cpu_inv = prepared.get("inverses_cpu")
        if fast and hasattr(model, "can_fuse_output_warp") and cpu_inv and all(r is not None for r in cpu_inv):
            theta_host = torch.cat(cpu_inv, dim=0)
            if model.can_fuse_output_warp(prepared["x"].shape, theta_host):
                warp_in_head = model.fuse_output_warp(torch.cat(inverses, dim=0), theta_host)
        fused_warp = not isinstance(warp_in_head, nullcontext)
        try:
            with warp_in_head:
                both = model(prepared["x"])
        finally:
            clear_noise(model)
        if isinstance(both, tuple):
            both = both[0]
        if fast:
            if not fused_warp:
                both = ops.affine_warp(both, torch.cat(inverses, dim=0), padding_mode="zeros", tta_grid_algebra=True)
            ta, tb = both[:nb], both[nb:]
            ta._dgtta_pair = tb._dgtta_pair = both
            ta._dgtta_guard_items = tb._dgtta_guard_items = per
            return ta, tb
        pieces = []
        for j in range(2 * steps):                  # general path: per (branch, step) slice, as calc_branch does
            t = both[j * per:(j + 1) * per]
            if not head_is_fused:
                t = map_label(t, get_map_idxs(label_mapping, optimized_labels, input_type="pretrain_labels"),
                              input_format="logits")
            t = after_mapping(t)
            if isinstance(t, tuple):
                t = t[0]
            if inverses[j] is not None:
                t = ops.affine_warp(t, inverses[j], padding_mode="zeros", tta_grid_algebra=True)
            pieces.append(t)
        targets = [torch.cat(pieces[:steps], dim=0) if steps > 1 else pieces[0],
                   torch.cat(pieces[steps:], dim=0) if steps > 1 else pieces[1]]
        targets[0]._dgtta_guard_items = targets[1]._dgtta_guard_items = per
    return targets[0], targets[1]


def _can_precompute_mind(model, modifier_fn_module):
    """MIND may be evaluated ahead of the network call only if nothing in front of mind_hook changes the input: the user's
    input modifier is the untouched template and the trainers' internal GIN augmentation is off (tta_main disables it)."""
    from ..gin import gin_hook
    from ..mind import mind_hook
    from ..utils import get_internal_augmentation_enabled
    hooks = list(model._forward_pre_hooks.values())
    if not hooks or hooks[-1] is not mind_hook or get_internal_augmentation_enabled():
        return False
    others = [h for h in hooks if h is not mind_hook and h is not gin_hook]
    mf = modifier_fn_module.ModifierFunctions
    return len(others) <= 1 and is_template_modifier(mf.modify_tta_input_fn, "modify_tta_input_fn")


def _pipeline_prep():
    import os
    return os.environ.get("DGTTA_PIPELINE_PREP", "1") != "0"


def batch_branches_enabled():
    """Both branches as one batch unless DGTTA_BATCH_BRANCHES=0 (then two calc_branch calls, as the reference does)."""
    import os
    return os.environ.get("DGTTA_BATCH_BRANCHES", "1") != "0"


def batched_steps(accum, batch_size):
    """Accumulation steps run per network pass (DGTTA_BATCH_STEPS, default 4): 2 * steps * batch_size samples must stay
    within the kernels' batch limits (loss: 8 pairs, warp: 16 samples) and `steps` must divide the number of accumulation steps."""
    import os
    k = max(1, int(os.environ.get("DGTTA_BATCH_STEPS", "4")))
    k = min(k, max(1, 8 // max(1, batch_size)), accum)
    while accum % k:
        k -= 1
    return k


def _fuse_head_if_possible(model, modifier_fn_module, label_mapping, optimized_labels):
    """map_label(logits) == evaluating only the mapped rows of the 1x1x1 head; valid iff the user's model-output
    modifier is the untouched template (identity)."""
    mf = modifier_fn_module.ModifierFunctions
    if hasattr(model, "set_selected_classes") and is_template_modifier(mf.modfify_tta_model_output_fn,
                                                                       "modfify_tta_model_output_fn"):
        model.set_selected_classes(get_map_idxs(label_mapping, optimized_labels, input_type="pretrain_labels"))
        return True
    return False


def tta_epoch(model, optimizer, config, tta_tens_list, patch_size, label_mapping, modifier_fn_module, device,
              head_is_fused, adapt):
    """ONE epoch of the unit loop (reference: tta.py:221-338): `patches_to_be_accumulated` steps of {get_batch, both
    branches, consistency loss, backward of loss/accum when `adapt`}, one AdamW step, then the evaluation patches.
    Returns (mean step loss, pseudo-Dice) as floats.  bench.py times exactly this function."""
    B = config["batch_size"]
    accum = config["patches_to_be_accumulated"]
    optimized_labels = config["optimized_labels"]
    intensity_aug_func = INTENSITY_AUG_FUNCTION_DICT[config["intensity_aug_function"]]
    # d(loss/accum), times the loss scale of the fp16 storage path (HipAdamW divides it out again and halves it after an
    # overflow: a step with inf / NaN gradients is skipped on the device and settled here, where the stream is drained)
    if adapt and hasattr(optimizer, "resolve_overflow") and optimizer.resolve_overflow():
        print(f"  gradient overflow in 16-bit storage: last optimizer step skipped, loss scale -> {optimizer.grad_scale:g}")
    scale = float(getattr(optimizer, "grad_scale", getattr(model, "loss_scale", 1.0)))
    inv_accum = torch.full((), scale / accum, dtype=torch.float32, device=device)
    model.train()
    step_losses = []

    def next_imgs():
        with torch.no_grad():
            imgs, _ = get_batch(tta_tens_list, numpy_rng().choice(range(len(tta_tens_list)), B).tolist(), patch_size,
                                fixed_patch_idx=None, device=device)
        return imgs[0] if len(imgs) == 1 else torch.cat(imgs, dim=0)

    if batch_branches_enabled():
        k = batched_steps(accum, B)
        n_pass = accum // k
        # The inputs of pass i+1 (patch sampling, GIN, affine warp, noise draws, MIND) do not depend on the weights or on
        # pass i: they are produced on a side stream while pass i is in its network passes (DGTTA_PIPELINE_PREP=0: in
        # line).  Draw order on both generators is unchanged - the draws happen when the work is enqueued.
        prep_stream = state_of(model).stream("prep_stream", device) if (_pipeline_prep() and torch.device(device).type == "cuda") \
            else None
        mind_ahead = prep_stream is not None and _can_precompute_mind(model, modifier_fn_module)
        main_stream = torch.cuda.current_stream(device) if prep_stream is not None else None

        def prepare(on_side):
            if not on_side:
                return prepare_both_branches(config, model, intensity_aug_func, B, next_imgs, device, steps=k,
                                             precompute_mind=mind_ahead)
            with torch.cuda.stream(prep_stream):
                pr = prepare_both_branches(config, model, intensity_aug_func, B, next_imgs, device, steps=k,
                                           precompute_mind=mind_ahead)
            for t in [pr["x"], pr["noise"], pr["feat"]] + list(pr["inverses"]):
                if torch.is_tensor(t):
                    t.record_stream(main_stream)       # consumed on the main stream: no reuse of the block before that
            return pr

        if prep_stream is not None:
            # once per epoch, before this epoch enqueues anything: the side stream sees the resident volumes (uploaded on
            # the main stream).  NOT per pass - that would order the next pass's inputs behind the current backward
            prep_stream.wait_stream(main_stream)
        prepared = prepare(False)
        for i_pass in range(n_pass):
            if prep_stream is not None:
                main_stream.wait_stream(prep_stream)
            target_a, target_b = run_both_branches(prepared, config, model, label_mapping, optimized_labels,
                                                   modifier_fn_module, head_is_fused, steps=k)
            prepared = None
            loss, dice = ops.consistency_loss(target_a, target_b, START_CLASS)      # mean of the k step losses
            if k == 1:
                step_losses.append(loss.detach())
            else:
                step_losses.extend((1.0 - dice.detach().reshape(k, B, -1)[:, :, START_CLASS:].mean((1, 2))).unbind(0))
            if adapt and loss.requires_grad:
                torch.autograd.backward(loss, grad_tensors=inv_accum * k)
            if i_pass + 1 < n_pass:
                prepared = prepare(prep_stream is not None)
    else:
        for _ in range(accum):
            imgs = next_imgs()
            args = (config, model, intensity_aug_func, None, patch_size, B, label_mapping, optimized_labels,
                    modifier_fn_module, imgs, device, head_is_fused)
            target_a = calc_branch("branch_a", *args)
            target_b = calc_branch("branch_b", *args)
            loss, _ = ops.consistency_loss(target_a, target_b, START_CLASS)
            step_losses.append(loss.detach())
            if adapt and loss.requires_grad:
                # d(loss/accum): the 1/accum factor is handed to the loss kernel as a device scalar
                torch.autograd.backward(loss, grad_tensors=inv_accum)
    if adapt:
        optimizer.step()
        optimizer.zero_grad()
    mean_loss_t = torch.stack(step_losses).mean()      # read back after the evaluation pass: one pipeline drain per epoch

    eval_dice = 0.0
    with torch.inference_mode():
        model.eval()
        for _ in range(config["tta_eval_patches"]):
            imgs, labels = get_batch(tta_tens_list, numpy_rng().choice(range(len(tta_tens_list)), B).tolist(),
                                     patch_size, fixed_patch_idx="center", device=device)
            keep = [i for i, l in enumerate(labels) if l is not None]
            if len(keep) == 0:
                eval_dice = float("nan")
                continue
            f_imgs = torch.cat([imgs[i] for i in keep], dim=0)
            f_labels = torch.cat([labels[i] for i in keep], dim=0)
            out = model(f_imgs)
            if isinstance(out, tuple):
                out = out[0]
            if not head_is_fused:
                out = map_label(out, get_map_idxs(label_mapping, optimized_labels, "pretrain_labels"), "logits")
            target_argmax, _ = ops.argmax_dice(out)
            f_labels = map_label(f_labels, get_map_idxs(label_mapping, optimized_labels, "tta_labels"),
                                 input_format="argmaxed").long()
            d = dice_coeff(target_argmax, f_labels, len(optimized_labels))
            eval_dice += 1 / config["tta_eval_patches"] * d.nanmean().item()
    return mean_loss_t.item(), eval_dice


def tta_unit(model, optimizer, config, tta_tens_list, patch_size, label_mapping, modifier_fn_module, device,
             head_is_fused, debug=False, progress=None):
    """Adapts `model` on one sample for config['epochs'] epochs (reference: tta.py:189-362). Returns (losses, dices)."""
    num_epochs, start = config["epochs"], config["start_tta_at_epoch"]
    tta_losses, eval_dices = torch.zeros(num_epochs), torch.zeros(num_epochs)
    model.apply(fix_all)
    for epoch in range(num_epochs):
        if epoch == start:
            model.apply(fix_all)
            mode = config["params_with_grad"]
            if mode == "all":
                model.apply(release_all)
            elif mode == "norms":
                model.apply(release_norms)
            elif mode == "encoder":
                model.encoder.apply(release_all)
            else:
                raise ValueError()
            n_released = sum(p.numel() for p in model.parameters() if p.requires_grad)
            print(f"Released #{n_released / 1e6:.2f} million trainable params")
        tta_losses[epoch], eval_dices[epoch] = tta_epoch(model, optimizer, config, tta_tens_list, patch_size,
                                                         label_mapping, modifier_fn_module, device, head_is_fused,
                                                         adapt=epoch >= start)
        if progress is not None:
            progress(epoch, float(tta_losses[epoch]), float(eval_dices[epoch]))
        if debug:
            break       # tta.py:334-335: debug runs stop after the first epoch
    return tta_losses, eval_dices


def tta_main(run_name, config, tta_data_dir, save_base_path, label_mapping, modifier_fn_module, device, debug=False,
             network_bundle=None, tta_data=None, shard=(0, 1), act_dtype=torch.float32, conv_impl=0):
    """Same signature as the reference's tta_main (tta.py:93-102) plus optional injection points:
    network_bundle=(predictor, patch_size, network, parameters) and tta_data=(iterable, num_samples) replace the
    nnU-Net loaders (used by tests / bench with synthetic data); shard=(rank, world) selects this process's units.

    A rank only ever loads / preprocesses the cases it works on (some ensemble member to adapt, or the case's ensemble
    prediction), and a case's tensors are dropped as soon as it is predicted: the reference keeps a second iterator over
    all cases for the inference loop (tta.py:149-155, 379-384); here a case whose members were all adapted by this rank is
    predicted right after its last member, the others after the filesystem barrier on the members' parameter files.  The
    prediction draws MIND noise from the device generator for every window batch (mind.py:150 does so too): the states of
    the CPU, device and numpy generators are saved around it and restored, so the TTA of the NEXT sample sees the draw
    sequence of the reference's order (all TTA first, then inference) - an externally seeded run without config['seed']
    follows the reference's stream."""
    from .nnunet_utils import load_network, load_tta_data
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dg_tta_amd runs on the MI355X only (device must be 'cuda' / 'cuda:N'); there is no CPU path")
    rank, world = shard
    save_path = Path(save_base_path) / run_name
    if network_bundle is None:
        try:
            network_bundle = load_network(config["pretrained_weights_filepath"], device, act_dtype=act_dtype,
                                          conv_impl=conv_impl)
        except BaseException:       # (a rank that cannot load the network: its peers stop instead of waiting for it)
            if world > 1 and save_path.parent.is_dir():
                save_path.mkdir(exist_ok=True, parents=False)
                mark_rank_failed(save_path, rank)
            raise
    predictor, patch_size, network, parameters = network_bundle
    across = config["tta_across_all_samples"]
    ensemble_count = config["ensemble_count"]
    num_epochs = config["epochs"]
    do_inference = config.get("run_inference", True) and not across and not debug

    def members_of(smp_idx, n_samples):
        return [e for e in range(ensemble_count) if unit_owner(smp_idx, e, n_samples, ensemble_count, world) == rank]

    def predicts(smp_idx, n_samples):
        return do_inference and unit_owner(smp_idx, 0, n_samples, ensemble_count, world) == rank

    timeout = float(config.get("barrier_timeout_s", 6 * 3600))
    results = {}
    try:
        # (everything from here on runs under the failure marker: a rank that dies while loading the data or writing the plan
        # must not leave its peers in the filesystem barrier until the timeout)
        save_path.mkdir(exist_ok=True, parents=False)
        for stale in (done_marker(save_path, rank), failed_marker(save_path, rank)):
            stale.unlink(missing_ok=True)       # (markers carry the launch id as well: peers ignore an earlier launch's)
        print("\n# Loading data")
        if tta_data is None:
            tta_data = load_tta_data(config, tta_data_dir, predictor, across,
                                     wanted=None if across else (lambda i, n: bool(members_of(i, n)) or predicts(i, n)))
        tta_data, num_samples = tta_data
        with open(save_path / "tta_plan.json", "w") as f:
            json.dump({k: v for k, v in config.items()}, f, indent=4)
        disable_internal_augmentation()
        n_units_samples = 1 if across else num_samples
        print("\n# Starting TTA")
        deferred = []
        for smp_idx in ([0] if across else range(num_samples)):
            sample, tta_tens_list, sample_id, _ext, sub_dir_tta = get_sample_specs(config, smp_idx, tta_data, save_path, across)
            mine = members_of(smp_idx, n_units_samples)
            predict_here = predicts(smp_idx, n_units_samples)
            if not mine and not predict_here:
                continue        # another GPU's case (independent: nothing to exchange); it was not loaded either
            if mine:
                print(f"\nSample {sample_id}")
            sub_dir_tta.mkdir(exist_ok=True, parents=True)
            for ensemble_idx in mine:
                ppath = get_parameters_save_path(sub_dir_tta, sample_id, ensemble_idx)
                if ppath.is_file():
                    print(f"TTA parameters file already exists. Skipping '{ppath}'")
                    continue
                if config.get("seed") is not None:      # optional, not a reference key: reproducible units on any GPU count
                    unit = get_global_idx([(smp_idx, max(num_samples, 1)), (ensemble_idx, ensemble_count)])
                    torch.manual_seed(int(config["seed"]) + unit)
                    np.random.seed(int(config["seed"]) + unit)
                model = get_model_from_network(network, modifier_fn_module, parameters).to(device)
                fused = _fuse_head_if_possible(model, modifier_fn_module, label_mapping, config["optimized_labels"])
                if hasattr(model, "accumulate_grads_in_place"):
                    model.accumulate_grads_in_place = True      # this loop owns the gradients (zero_grad once per epoch)
                    model.exact_zero_bias_grad = True           # see HipPlainConvUNet.exact_zero_bias_grad
                optimizer = HipAdamW(model.parameters(), lr=config["lr"], grad_scale=getattr(model, "loss_scale", 1.0))

                def progress(epoch, loss, dice):
                    print(f"  epoch {epoch}: loss={loss:.3f}, Pseudo-Dice={dice * 100:.1f}%", flush=True)

                losses, dices = tta_unit(model, optimizer, config, tta_tens_list, patch_size, label_mapping,
                                         modifier_fn_module, device, fused, debug, progress)
                if fused:
                    model.set_selected_classes(None)
                # written under a temporary name and renamed: the owner of the sample polls for the file (filesystem barrier)
                tmp = ppath.with_name(ppath.name + f".tmp{rank}")
                torch.save([model.state_dict()], tmp)
                tmp.replace(ppath)
                results[(sample_id, ensemble_idx)] = (losses, dices)
                if num_epochs > 0:
                    plot_run_results(sub_dir_tta, sample_id, ensemble_idx, losses, dices)
                if debug:
                    break
            release_resident(tta_tens_list)      # this sample's volume leaves HBM
            if predict_here:
                case = (sample, sample_id, sub_dir_tta)
                if len(mine) == ensemble_count:
                    np_state = np.random.get_state()
                    with torch.random.fork_rng(devices=[device]):      # CPU + device generator states restored on exit
                        _predict_case(case, config, network, predictor, patch_size, label_mapping, modifier_fn_module, device,
                                      save_path, tta_data_dir, results, world, timeout)
                    np.random.set_state(np_state)
                else:
                    deferred.append(case)            # members adapted on other GPUs: predicted after this rank's units
        # ---- ensemble sliding-window inference with the TTA'd parameter sets (reference: tta.py:376-416)
        if do_inference:
            if deferred:
                print("\n\n# Starting inference")
            for case in deferred:
                _predict_case(case, config, network, predictor, patch_size, label_mapping, modifier_fn_module, device,
                              save_path, tta_data_dir, results, world, timeout)
            del deferred
            # ---- evaluation over the whole run directory: only once every rank's predictions are on disk
            mark_rank_done(save_path, rank)
            if world == 1:
                results.update(evaluate_run(save_path, config, modifier_fn_module, device))
            elif rank == 0 and not summary_by_parent():
                wait_for_done_markers(save_path, world, timeout)
                results.update(evaluate_run(save_path, config, modifier_fn_module, device))
    except BaseException:
        if world > 1:           # peers waiting in the filesystem barrier stop instead of running into its timeout
            mark_rank_failed(save_path, rank)
        raise
    return results


def _predict_case(case, config, network, predictor, patch_size, label_mapping, modifier_fn_module, device, save_path,
                  tta_data_dir, results, world, timeout):
    """Ensemble prediction of one case + its evaluation target (reference: tta.py:379-446).

    NIfTI cases are exported the way nnU-Net's predictor does it for the reference (nnunet_utils.py:208-230): the
    ensemble logits are resampled to the shape the case had before preprocessing, argmax'd, pasted into the crop box and
    transposed back, then written with the image's own header - the prediction is interchangeable with a reference run's
    and lines up with `labels{Ts,Tr}/<case>`, which is what it is evaluated against (tta.py:420-447).  Array cases
    (.npy/.npz/.pt: already preprocessed, no geometry) are written as they are."""
    from .inference import export_segmentation, predict_ensemble
    from .image_io import read_image as read_nifti
    from .torch_utils import get_imgs
    sample, sample_id, sub_dir_tta = case
    ensemble_count = config["ensemble_count"]
    optimized_labels = config["optimized_labels"]
    paths = [get_parameters_save_path(sub_dir_tta, sample_id, e) for e in range(ensemble_count)]
    if world > 1:       # members adapted on other GPUs: wait for their files (no collective)
        wait_for_files(paths, timeout, abort_if=[failed_marker(save_path, r) for r in range(world)])
    if not all(p.is_file() for p in paths):
        return
    print(f"\nPredicting {sample_id}")
    params = [torch.load(p, map_location=device)[0] for p in paths]
    model = get_model_from_network(network, modifier_fn_module).to(device)
    disable_internal_augmentation()
    image = get_imgs(sample["data"].unsqueeze(0)).squeeze(0)
    props = sample.get("data_properties") or {}
    nii = props.get("nifti_header")
    acc, nsum, crop = predict_ensemble(image, model, params, patch_size)
    plans = getattr(predictor, "plans", None)
    original = nii is not None and plans is not None and "shape_before_cropping" in props
    seg = export_segmentation(acc, nsum, crop, props if original else None, plans, getattr(predictor, "configuration", None))
    del acc, nsum
    seg = map_label(torch.as_tensor(seg.astype(np.int64))[None],
                    get_map_idxs(label_mapping, optimized_labels, "pretrain_labels"), input_format="argmaxed")[0]
    # (the prediction is written in the format of the case's image, as SimpleITK's writer picks it from the file name)
    out = Path(str(save_path / sample_id) + ((nii.get("ext") or ".nii.gz") if nii is not None else ".npy"))
    out.parent.mkdir(exist_ok=True, parents=True)
    _save_label_map(out, seg.numpy().astype(np.int16), nii)
    results[(sample_id, "prediction")] = out
    bucket = "Ts" if "outputTs" in out.parent.name else ("Tr" if "outputTr" in out.parent.name else None)
    if bucket is None:
        return
    tta_idxs = get_map_idxs(label_mapping, optimized_labels, "tta_labels")
    ref_path = save_path / f"mapped_target_labels{bucket}" / out.name
    orig_target = Path(tta_data_dir) / f"labels{bucket}" / out.name if tta_data_dir is not None else None
    if original and orig_target is not None and orig_target.is_file():
        # the untouched label file of the case, mapped into the index space of optimized_labels (tta.py:431-446)
        tgt, tgt_hdr = read_nifti(orig_target)
        tgt = map_label(torch.as_tensor(tgt.astype(np.int64))[None], tta_idxs, input_format="argmaxed")[0]
        ref_path.parent.mkdir(exist_ok=True, parents=True)
        _save_label_map(ref_path, tgt.numpy().astype(np.int16), tgt_hdr)
    elif sample["data"].shape[0] > 1 and not original:
        # array cases: the label channels of the preprocessed sample are the only reference there is
        segs = sample["data"][1:]
        target = torch.cat([(segs.sum(0, keepdim=True) < 1.0).float(), segs.float()], dim=0).argmax(0)
        target = map_label(target[None], tta_idxs, input_format="argmaxed")[0]
        ref_path.parent.mkdir(exist_ok=True, parents=True)
        _save_label_map(ref_path, target.numpy().astype(np.int16), nii)


def evaluate_run(save_path, config, modifier_fn_module, device="cuda"):
    """postprocess_results_fn + summary_{Ts,Tr}.json over a finished run directory (reference: tta.py:447-470).  Runs in
    the single process, in rank 0 after the filesystem barrier, or in the fan-out parent of `dgtta run_tta --gpus N`
    after every child has exited."""
    from .evaluation import compute_metrics_on_folder_simple
    save_path, out = Path(save_path), {}
    for bucket in ["Ts", "Tr"]:
        refs, preds = save_path / f"mapped_target_labels{bucket}", save_path / f"tta_output{bucket}"
        if refs.is_dir() and preds.is_dir():
            modifier_fn_module.ModifierFunctions.postprocess_results_fn(preds)
            summary = compute_metrics_on_folder_simple(refs, preds, list(range(len(config["optimized_labels"]))),
                                                       output_file=save_path / f"summary_{bucket}.json", device=device)
            out[("summary", bucket)] = summary["foreground_mean"]["Dice"]
    return out


def _save_label_map(path, arr, nifti_header=None):
    if str(path).endswith(".npy"):
        np.save(path, arr)
    else:
        from .image_io import write_image
        write_image(path, arr, header=nifti_header)
