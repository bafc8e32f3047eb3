"""Sliding-window inference timing (SURVEY config 3 scaled down): infbench.py <volume edge> <window batch>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dg_tta_amd.unet import HipPlainConvUNet
from dg_tta_amd.tta import inference
n, wb = int(sys.argv[1]), int(sys.argv[2])
inference.WINDOW_BATCH = wb
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = HipPlainConvUNet(act_dtype=torch.bfloat16).to(dev)
from oracle.unet import init_he          # scratch tool only: seeded He init as the tests / bench use
init_he(net, seed=0)
data = torch.randn(12, n, n, n).to(dev)      # resident, as run_inference keeps it across ensemble members
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc, nsum, crop = inference.predict_sliding_window_return_logits(net, data, [128, 128, 128])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
nw = len(inference.compute_steps_for_sliding_window((n, n, n), [128] * 3, 0.5)[0]) ** 3
print(f"{n}^3 volume, {nw} windows, window batch {wb}: {dt:.3f} s = {dt / nw * 1e3:.2f} ms per window; checksum {float(acc.sum()):.6e}")
# breakdown: forward only / accumulate only (synchronised)
work = data[None, :, :128, :128, :128].contiguous().to(dev)
net.eval()
with torch.no_grad():
    for _ in range(2): out = net(work)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = net(work)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"forward only: {(t1 - t0) / 10 * 1e3:.2f} ms per window, out {tuple(out.shape)} {out.dtype} cl3d={out.is_contiguous(memory_format=torch.channels_last_3d)}")
    t0 = time.perf_counter()
    for _ in range(10): o2 = out.float().contiguous(memory_format=torch.channels_last_3d)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"float+contiguous: {(t1 - t0) / 10 * 1e3:.2f} ms")
