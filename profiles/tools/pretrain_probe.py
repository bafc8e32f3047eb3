"""How fast does the full 3d_fullres net learn the synthetic atlas task through the engine?  (bench.py's in-bench
pre-training: picks steps / lr.)  usage: pretrain_probe.py [steps] [lr] [batch] [dtype] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dg_tta_amd.mind import MIND3D
from dg_tta_amd.pretraining.hooks import register_dg_hooks
from dg_tta_amd.pretraining.supervised import pretrain_supervised
from dg_tta_amd.synthetic import atlas_case, he_init_, synthetic_label_mapping
from dg_tta_amd.tta.torch_utils import dice_coeff, get_batch
from dg_tta_amd.unet import HipPlainConvUNet
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-3
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[4] if len(sys.argv) > 4 else "fp16"]
ncases = int(sys.argv[5]) if len(sys.argv) > 5 else 6
DEV, S, P, K = "cuda:0", 160, [128] * 3, 15
t0 = time.time()
cases = [atlas_case(S, K, s, "source") for s in range(ncases)]
print(f"{ncases} source cases: {time.time() - t0:.1f} s", flush=True)
mapping, names = synthetic_label_mapping(K)
SEL = os.environ.get("PROBE_SEL", "0") == "1"          # train the C_opt selected head rows only (labels = positions)
lut = torch.arange(K + 1) if SEL else torch.tensor([mapping[n][0] for n in names])
net = he_init_(HipPlainConvUNet(act_dtype=dt), seed=7)
handles = register_dg_hooks(net)
net = net.to(DEV)
if SEL:
    net.set_selected_classes(torch.tensor([mapping[n][0] for n in names]))
from dg_tta_amd.optim import HipAdamW
opt = HipAdamW(list(net.parameters()), lr=lr, weight_decay=0.0, grad_scale=net.loss_scale)
PS = int(os.environ.get("PROBE_PATCH", "128"))
PT = [PS] * 3
torch.manual_seed(5); torch.cuda.manual_seed(5); np.random.seed(5)

def evaluate(tag):
    res = {}
    for dom, seed in (("source", 77), ("target", 31)):
        c = atlas_case(S, K, seed, dom)
        with torch.no_grad():
            net.eval()
            imgs, labels = get_batch([c], [0], P, "center", DEV)
            out = net.forward(MIND3D()(imgs[0], out_dtype=dt))
            d = dice_coeff(out.argmax(1, keepdim=True), lut.to(DEV)[labels[0]], K + 1 if SEL else 105)
            net.train()
        res[dom] = float(d.nanmean()) if SEL else float(d[[3 * i - 1 for i in range(1, K + 1)]].nanmean())
    print(f"{tag}: hard Dice source {res['source']:.4f}, target {res['target']:.4f}", flush=True)

# PROBE_PHASE2=<n>: after `steps` steps on the selected classes, n more steps over ALL 105 classes (labels at the pretrain ids)
phase2 = int(os.environ.get("PROBE_PHASE2", "0"))
done = 0
for chunk in range(0, steps, 50):
    n = min(50, steps - chunk)
    torch.cuda.synchronize(); t0 = time.time()
    losses = pretrain_supervised(net, cases, PT, lut, steps=n, batch=batch, lr=lr, device=DEV, optimizer=opt)
    torch.cuda.synchronize(); dt_s = time.time() - t0
    done += n
    print(f"steps {done}: loss {float(losses[:5].mean()):.3f} -> {float(losses[-5:].mean()):.3f}, {dt_s / n * 1e3:.1f} ms / step", flush=True)
    evaluate(f"after {done}")

if phase2:
    sel_ids = torch.tensor([mapping[n][0] for n in names])
    net.set_selected_classes(None)
    SEL = False
    lut = sel_ids
    opt = HipAdamW(list(net.parameters()), lr=lr, weight_decay=0.0, grad_scale=net.loss_scale)

    def evaluate2(tag):
        res = {}
        for dom, seed in (("source", 77), ("target", 31)):
            c = atlas_case(S, K, seed, dom)
            with torch.no_grad():
                net.eval()
                net.set_selected_classes(sel_ids)
                imgs, labels = get_batch([c], [0], P, "center", DEV)
                out = net.forward(MIND3D()(imgs[0], out_dtype=dt))
                d = dice_coeff(out.argmax(1, keepdim=True), labels[0], K + 1)
                alive = float((out.float().sum(1) > 0).float().mean())
                net.set_selected_classes(None)
                net.train()
            res[dom] = (float(d.nanmean()), alive)
        print(f"{tag}: hard Dice source {res['source'][0]:.4f}, target {res['target'][0]:.4f}; mapped logit sum > 0 on {res['target'][1]:.4f} of the target voxels", flush=True)
    evaluate2("phase 2 start")
    for chunk in range(0, phase2, 20):
        n = min(20, phase2 - chunk)
        torch.cuda.synchronize(); t0 = time.time()
        losses = pretrain_supervised(net, cases, PT, lut, steps=n, batch=batch, lr=lr, device=DEV, optimizer=opt)
        torch.cuda.synchronize()
        print(f"phase 2 steps {chunk + n}: loss {float(losses[:3].mean()):.3f} -> {float(losses[-3:].mean()):.3f}, {(time.time() - t0) / n * 1e3:.1f} ms / step", flush=True)
        evaluate2(f"phase 2 after {chunk + n}")
