"""BASELINE config 1 on the GPU: the whole `dgtta prepare_tta` + `dgtta run_tta` command line on ONE synthetic 64^3 "MRI"
case stored as NIfTI (raw-case preprocessing included), full nnUNet 3d_fullres topology (seeded He-init stand-in for the
TS104_GIN_MIND checkpoint, which cannot be downloaded), 1 TTA epoch with backward, ensemble inference and summary."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prepare_and_run_tta_cli_full_net_64(tmp_path, monkeypatch):
    from dg_tta_amd.run import DGTTAProgram
    from dg_tta_amd.synthetic import he_init_, synthetic_case
    from dg_tta_amd.tta.nifti_io import read_nifti, write_nifti
    from dg_tta_amd.unet import HipPlainConvUNet
    raw = tmp_path / "raw" / "Dataset803_Target"
    (raw / "imagesTs").mkdir(parents=True)
    (raw / "labelsTs").mkdir()
    case = synthetic_case(size=64, k=3, seed=5)                      # [1+3, 64,64,64]: image + 3 ellipsoid labels
    img = (case[0] * 300.0).numpy()                                  # CT-like range for the plans' CTNormalization
    lab = torch.cat([(case[1:].sum(0, keepdim=True) < 1).float(), case[1:]]).argmax(0).numpy().astype(np.int16)
    write_nifti(raw / "imagesTs" / "mr01_0000.nii.gz", img.astype(np.float32), spacing=(1.5, 1.5, 1.5))
    write_nifti(raw / "labelsTs" / "mr01.nii.gz", lab, spacing=(1.5, 1.5, 1.5))
    # second case: anisotropic (2.5 x 1.1 x 1.1 mm), with a zero border that preprocessing crops away -> its prediction has to
    # be resampled back, un-cropped and written in THIS geometry
    img2 = np.zeros((40, 76, 74), np.float32)
    lab2 = np.zeros((40, 76, 74), np.int16)
    img2[3:37, 6:70, 5:69] = img[:34] + 700.0
    lab2[3:37, 6:70, 5:69] = lab[:34]
    write_nifti(raw / "imagesTs" / "mr02_0000.nii.gz", img2, spacing=(1.1, 1.1, 2.5))         # (x, y, z)
    write_nifti(raw / "labelsTs" / "mr02.nii.gz", lab2, spacing=(1.1, 1.1, 2.5))
    json.dump({"labels": {"background": 0, "liver": 1, "spleen": 2, "my_organ": 3}}, open(raw / "dataset.json", "w"))
    root = tmp_path / "dgroot"
    root.mkdir()
    for k, v in {"nnUNet_raw": str(tmp_path / "raw"), "nnUNet_results": str(tmp_path / "res"),
                 "nnUNet_preprocessed": str(tmp_path / "pre"), "DG_TTA_ROOT": str(root)}.items():
        monkeypatch.setenv(k, v)
    DGTTAProgram(["dgtta", "prepare_tta", "TS104_GIN_MIND", "803"])
    plan_dir = root / "plans" / "Pretrained_TS104_GIN_MIND_at_Dataset803_Target" / "nnUNetTrainer_GIN_MIND__3d_fullres" / "fold_0"
    plan = json.load(open(plan_dir / "tta_plan.json"))
    assert sorted(Path(p).name for p in plan["tta_data_filepaths"]) == ["mr01_0000.nii.gz", "mr02_0000.nii.gz"]
    # the model folder: patch 64^3 (config 1), seeded stand-in checkpoint with the real key layout
    weights = Path(plan["pretrained_weights_filepath"])
    mplans = json.load(open(weights.parents[1] / "plans.json"))
    mplans["configurations"]["3d_fullres"]["patch_size"] = [64, 64, 64]
    json.dump(mplans, open(weights.parents[1] / "plans.json", "w"))
    net = he_init_(HipPlainConvUNet(), seed=7)
    torch.save({"network_weights": net.state_dict(), "trainer_name": "nnUNetTrainer_GIN_MIND"}, weights)
    # 1 epoch that adapts (start_tta_at_epoch = 0), 4 accumulation steps, one ensemble member
    plan.update(epochs=1, start_tta_at_epoch=0, patches_to_be_accumulated=4, ensemble_count=1)
    json.dump(plan, open(plan_dir / "tta_plan.json", "w"), indent=4)
    DGTTAProgram(["dgtta", "run_tta", "TS104_GIN_MIND", "803", "--device", "cuda:0"])        # default storage: fp16
    runs = sorted((root / "results" / "Pretrained_TS104_GIN_MIND_at_Dataset803_Target" /
                   "nnUNetTrainer_GIN_MIND__3d_fullres" / "fold_0").iterdir())
    assert len(runs) == 1
    run = runs[0]
    out = run / "tta_outputTs"
    assert (run / "tta_plan.json").is_file() and (out / "mr01__ensemble_idx_0_tta_parameters.pt").is_file()
    state = torch.load(out / "mr01__ensemble_idx_0_tta_parameters.pt", map_location="cpu")[0]
    assert set(state) == set(net.state_dict())
    moved = sum(int(not torch.equal(state[k], v)) for k, v in net.state_dict().items())
    assert moved > 30                                              # the adaptation step changed the parameters
    seg, hdr = read_nifti(out / "mr01.nii.gz")                     # prediction written with the case's geometry
    assert seg.shape == (64, 64, 64) and hdr["pixdim"] == pytest.approx((1.5, 1.5, 1.5))
    assert set(np.unique(seg).tolist()) <= {0, 1, 2}               # ids of the optimized labels (background, liver, spleen)
    tgt, _ = read_nifti(run / "mapped_target_labelsTs" / "mr01.nii.gz")
    assert (tgt == 1).sum() == (lab == 1).sum() and (tgt == 2).sum() == (lab == 2).sum() and (tgt == 3).sum() == 0
    sj = json.loads((run / "summary_Ts.json").read_text())
    assert len(sj["metric_per_case"]) == 2 and set(sj["mean"]) == {"0", "1", "2"}
    # the anisotropic, cropped case comes back in ITS OWN geometry and is evaluated against the untouched label file
    seg2, hdr2 = read_nifti(out / "mr02.nii.gz")
    assert seg2.shape == (40, 76, 74) and hdr2["pixdim"] == pytest.approx((1.1, 1.1, 2.5))
    outside = np.ones(seg2.shape, bool)
    outside[3:37, 6:70, 5:69] = False
    assert (seg2[outside] == 0).all() and (seg2 != 0).any()        # zeros outside the crop box, labels inside
    tgt2, thdr2 = read_nifti(run / "mapped_target_labelsTs" / "mr02.nii.gz")
    assert tgt2.shape == lab2.shape and thdr2["pixdim"] == pytest.approx((1.1, 1.1, 2.5))
    assert np.array_equal(tgt2, np.where(lab2 == 3, 0, lab2))      # my_organ is not optimised: mapped to background
    m2 = [c for c in sj["metric_per_case"] if c["prediction_file"].endswith("mr02.nii.gz")][0]["metrics"]
    assert m2["1"]["n_ref"] == int((lab2 == 1).sum()) and m2["1"]["n_pred"] == int((seg2 == 1).sum())


def test_bench_two_real_ranks_on_one_gpu():
    """bench.py --gpus 2 end to end with the REAL runner: the launcher starts two fresh rank
    processes, each adapts its own sample with the product's tta_epoch, they meet at the barrier (gloo here, because both share
    this box's one GPU - `--share-gpu`; on an 8-GPU node the same code path runs one rank per GPU over RCCL), and rank 0 prints
    one line for the whole job."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1",
                          "--size", "64", "--accum", "4", "--pretrain-steps", "30", "--cpu-size", "64", "--inference-size", "128"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and len(out["per_rank_epochs_per_s"]) == 2
    assert out["value"] == pytest.approx(2 * out["value_per_gpu"], rel=1e-4)
    assert out["value_per_gpu"] <= min(out["per_rank_epochs_per_s"]) * 1.0001
    assert 0.0 < out["loss_last_epoch"] < 1.0 and out["roofline"] is not None
    # the fp32 / oracle-refereed legs are single-GPU work; the CPU baseline rides on rank 0 after the timed region (round 5)
    assert "fp32" not in out and out["cpu_baseline"]["value"] > 0 and "pretraining" in out and "epoch_roofline" in out
    # ... and so do the sliding-window leg and the at-size parity record (rank 0, peers in the final barrier)
    assert out["inference"]["windows"] == 27 and out["inference"]["fp32_logits_accumulator"]["label_agreement_with_feature_accumulator"] > 0.999
    assert "parity_at_size" in out


def test_bench_four_real_ranks_rehearsal_and_rank0_matches_the_single_run():
    """rehearsal of the hardware SCALE run so that it cannot fail on plumbing - `bench.py --gpus N --share-gpu`
    with the REAL runner at 64^3.  N = 4 here: the pool admits at most 6 processes on a box's GPU and this test process is one
    of them (8 ranks run with the stub runner on the CPU: tests/test_sharding_gloo.py).  The line carries one rate per rank,
    the world size the rendezvous saw, the CPU baseline and the roofline object on rank 0; rank 0's sample, seeds and draws
    do not depend on the number of ranks: its trajectory is the N = 1 run's, bit for bit."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--size", "64", "--accum", "4", "--pretrain-steps", "30", "--no-fp32",
              "--inference-size", "0", "--cpu-size", "64", "--cpu-warmup", "0", "--no-parity"]

    def line(extra):
        res = subprocess.run([sys.executable, str(root / "bench.py")] + extra + common, env=env, capture_output=True, text=True,
                             timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])
    four, one = line(["--gpus", "4", "--share-gpu"]), line(["--gpus", "1"])
    assert four["n_gpus"] == 4 and len(four["per_rank_epochs_per_s"]) == 4 and four["config"]["world_size_seen"] == 4
    assert four["scaling"] == "weak" and four["value"] == pytest.approx(4 * four["value_per_gpu"], rel=1e-4)
    assert four["cpu_baseline"]["value"] > 0 and four["cpu_baseline"]["kind"] == "port"
    assert four["roofline"] is not None and "epoch_frac" in four["roofline"] and "largest_consumer" in four["roofline"]
    assert four["dtype"] == one["dtype"] == "fp16"
    assert four["loss_last_epoch"] == one["loss_last_epoch"] and four["pseudo_dice"] == one["pseudo_dice"]
