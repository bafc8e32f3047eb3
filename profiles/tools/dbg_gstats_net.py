import os, sys, torch
sys.path.insert(0, '.')
from dg_tta_amd import _lib
from dg_tta_amd.synthetic import he_init_
from dg_tta_amd.unet import HipPlainConvUNet
DEV = "cuda:0"
torch.manual_seed(4)
net = he_init_(HipPlainConvUNet(act_dtype=torch.float16), seed=7)
for m in net.modules():
    if m.__class__.__name__ == "HipInstanceNorm3d":
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
net = net.to(DEV)
net.set_selected_classes(torch.arange(16) * 3)
x = torch.rand(2, 12, 64, 64, 64, device=DEV)
gout = torch.randn(2, 16, 64, 64, 64, device=DEV).contiguous(memory_format=torch.channels_last_3d)
def run(flag):
    os.environ["DGTTA_IN_GSTATS"] = flag
    _lib.load().dgtta_reload_env()
    net.zero_grad(); net(x).backward(gout); torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
g1, g0 = run("1"), run("0")
for n in g0:
    sc = float(g0[n].abs().max())
    err = float((g1[n] - g0[n]).abs().max())
    if err > 1e-4 * sc + 1e-12:
        print(f"{n:50s} scale {sc:.3e} err {err:.3e}")
