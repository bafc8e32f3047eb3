"""Time of the export in the original geometry on a realistic CT: logits on the 1.5 mm grid -> 0.8 mm original grid."""
import os, sys, time, json, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dg_tta_amd.tta.inference import export_segmentation
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
PLANS = json.loads((ROOT / "dg_tta_amd" / "__resources__" / "model_skeleton" / "plans.json").read_text())
DEV = "cuda:0"
pre = tuple(int(v) for v in (sys.argv[1:4] or (213, 273, 273)))
orig = tuple(int(v) for v in (sys.argv[4:7] or (400, 512, 512)))
C = 105
g = torch.Generator(device=DEV).manual_seed(0)
acc = torch.randn(*pre, C, device=DEV, generator=g)
nsum = torch.rand(*pre, device=DEV, generator=g) + 0.5
props = {"shape_before_cropping": orig, "bbox_used_for_cropping": [[0, s] for s in orig],
         "shape_after_cropping_and_before_resampling": orig, "spacing": [0.8, 0.8, 0.8]}
crop = [slice(0, s) for s in pre]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg = export_segmentation(acc, nsum, crop, props, PLANS, "3d_fullres")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"export {pre} -> {orig}, {C} classes: {dt:.2f} s, labels {len(np.unique(seg[::8, ::8, ::8]))}", flush=True)
