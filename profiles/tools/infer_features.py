"""Feature-space sliding-window inference of one member at 512^3, timed three times per setting:
python3 profiles/tools/infer_features.py  (DGTTA_FEATURE_FOLD=0/1 in the environment selects the folded apply)"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from dg_tta_amd.mind import mind_hook
from dg_tta_amd.synthetic import he_init_
from dg_tta_amd.tta.inference import accumulate_window_features
from dg_tta_amd.unet import HipPlainConvUNet

n, dev = 512, torch.device("cuda:0")
net = he_init_(HipPlainConvUNet(act_dtype=torch.float16 if "fp16" in sys.argv else torch.bfloat16), seed=7)
net.register_forward_pre_hook(mind_hook)
net = net.to(dev)
vol = torch.randn(1, n, n, n, generator=torch.Generator().manual_seed(3)).to(dev)
patch = [128] * 3
accumulate_window_features(net, vol[:, :128, :128, :128], patch)
facc = torch.zeros((n, n, n, 32), dtype=torch.float32, device=dev)
for rep in range(3):
    facc.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    accumulate_window_features(net, vol, patch, facc)
    torch.cuda.synchronize()
    print(f"rep {rep}: {time.perf_counter() - t0:.4f} s", flush=True)
