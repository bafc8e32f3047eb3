"""The PRODUCT's TTA run against the CPU ORACLE's, every storage type, with the stated tolerances.

What `bench.py`'s `dice_delta` reports in every default run, asserted on the driver's box: the full 3d_fullres net is
pre-trained through the engine on the source domain of the synthetic atlas task (550 steps at 64^3, ~6 s), then
`oracle/tta.py:tta_unit` - the restatement of `dg_tta/tta/tta.py:189-340` that tests/golden/make_golden_r2.py / _r5.py pin
bit for bit against the reference's own loop - adapts it on the host for 3 epochs x 8 accumulation steps on 64^3 patches of a
target-domain volume (AdamW, lr 3e-4: two optimizer steps), and the product's `tta_unit` runs the same draw stream
(oracle/replay.py) in fp32, fp16 and bf16 storage.

Stated tolerances (bench.py): hard Dice vs ground truth and pseudo-Dice within 1e-3 for EVERY storage type
(north_star); per-epoch soft-Dice loss within 1e-5 for fp32 (and the label map identical wherever the oracle's top-2 margin
exceeds 1e-3), within 1e-3 + 2 x the fp32 drift for 16-bit storage.  fp16 - the default 16-bit type - meets all of them.
bf16 meets the Dice clause; its loss is NOT claimed (~1e-3 on unchanged weights: the reference's hard mask `sum_c logits > 0`
amplifies the 8-bit significand's forward error) and only fenced by a regression bound here."""
import json
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def referee():
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import bench
    args = bench.parse_args(["--referee-patch", "64", "--referee-epochs", "3", "--referee-accum", "8"])
    rec = bench.referee_tta_run(args, DEV)
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "referee_3_epochs.json").write_text(json.dumps(rec, indent=1))
    return bench, rec


def test_the_referee_run_adapts_a_model_that_segments(referee):
    _, rec = referee
    assert rec["epochs"] == 3 and rec["accum"] == 8 and rec["patch"] == 64
    o = rec["oracle"]
    assert o["hard_dice_vs_gt_before"] > 0.85                       # a Dice delta on a model that segments nothing says nothing
    assert 0.05 < o["loss_per_epoch"][0] < 0.9                      # the consistency mask is alive
    assert o["loss_per_epoch"][2] != o["loss_per_epoch"][1]         # optimizer steps moved the model


def test_fp32_reproduces_the_oracle_run(referee):
    bench, rec = referee
    e = rec["fp32"]
    assert e["loss"] <= bench.FP32_LOSS_TOLERANCE, e
    assert e["pseudo_dice"] <= 5e-5 and e["hard_dice"] <= 1e-6, e
    assert e["label_agreement_where_margin_gt_1e-3"] == 1.0 and e["label_agreement"] >= 0.99999, e
    assert e["skipped_optimizer_steps"] == 0 and e["within_tolerance"]


def test_fp16_storage_is_within_the_stated_tolerances(referee):
    bench, rec = referee
    e = rec["fp16"]
    assert e["loss_tolerance"] == pytest.approx(bench.LOSS_TOLERANCE_16BIT + 2 * rec["fp32_drift"])
    assert e["loss"] <= e["loss_tolerance"], e
    assert e["pseudo_dice"] <= bench.DICE_TOLERANCE and e["hard_dice"] <= bench.DICE_TOLERANCE, e
    assert e["hard_dice_per_class_max"] <= bench.DICE_TOLERANCE, e          # (stronger than the mean the clause speaks of)
    assert e["label_agreement_where_margin_gt_1e-3"] >= 0.9999 and e["skipped_optimizer_steps"] == 0, e
    assert e["within_tolerance"]
    # the Dice quantities with a factor of three to spare (measured 3.7e-5 / 4.2e-5).  The per-epoch LOSS has no such fence: it
    # depends on the pre-trained instance through the reference's hard mask `sum_c logits > 0` - two instances of the same
    # recipe (the small-plane weight gradients changed their summation order between them, so the 550 pre-training steps ended
    # in different weights) gave 2.7e-5 and 6.7e-4 at three epochs, 2.8e-5 and 1.1e-4 on the unchanged weights of epoch 0
    assert e["pseudo_dice"] <= 3e-4 and e["hard_dice"] <= 3e-4, e


def test_bf16_storage_holds_the_dice_clause(referee):
    bench, rec = referee
    e = rec["bf16"]
    assert e["pseudo_dice"] <= bench.DICE_TOLERANCE and e["hard_dice"] <= bench.DICE_TOLERANCE and e["dice_within_tolerance"], e
    assert e["label_agreement"] >= 0.999, e
    # NOT a parity claim: a fence around the measured ~1.1e-3 so that a 3x regression is noticed
    assert e["loss"] <= 3.5e-3, e
