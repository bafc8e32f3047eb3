"""One Conv-IN-LReLU block (deep-layer shape) on the HIP fp32 path vs float64, next to torch-CPU fp32 vs float64."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from dg_tta_amd.unet import HipPlainConvUNet
from oracle import unet as ounet

def run(cin, cout, n, impl, seed=0):
    cfg = dict(features=(cout,), strides=(1,), n_conv_enc=(1,), n_conv_dec=(), in_channels=cin, num_classes=cout)
    torch.manual_seed(seed)
    om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(cfg), 3), 4)
    return om

# the product net needs >= 2 stages; use a 2-stage net whose first stage is the block under test
def nets(c0, c1, impl):
    cfg = dict(features=(c0, c1), strides=(1, 2), n_conv_enc=(2, 2), n_conv_dec=(2,), in_channels=12, num_classes=5)
    om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(cfg), 3), 4)
    hm = HipPlainConvUNet(cfg, conv_impl=impl)
    hm.load_state_dict(om.state_dict())
    return cfg, om, hm.to("cuda:0")

for (c0, c1, n) in ((256, 320, 8), (128, 256, 16), (32, 64, 32)):
    for impl in (0, 1):
        cfg, om, hm = nets(c0, c1, impl)
        torch.manual_seed(1)
        x = torch.randn(1, 12, n, n, n)
        w = torch.randn(1, 5, n, n, n)
        o64 = ounet.PlainConvUNetOracle(cfg).double(); o64.load_state_dict(om.state_dict())
        y64 = o64(x.double()); (y64 * w.double()).sum().backward()
        y32 = om(x); (y32 * w).sum().backward()
        yh = hm(x.to("cuda:0")); (yh * w.to("cuda:0")).sum().backward()
        print(f"features ({c0},{c1}) at {n}^3 impl {impl}: out err hip {((yh.detach().cpu().double()-y64).abs().max()/y64.abs().max()).item():.2e} torch32 {((y32.double()-y64).abs().max()/y64.abs().max()).item():.2e}")
        g64 = dict(o64.named_parameters()); g32 = dict(om.named_parameters()); gh = dict(hm.named_parameters())
        for name in g64:
            if g64[name].grad is None or ".all_modules." in name or name.startswith("decoder.encoder"):
                continue
            if name.endswith("conv.bias") and ".convs." in name: continue
            s = g64[name].grad.abs().max().item()
            e_h = (gh[name].grad.cpu().double() - g64[name].grad).abs().max().item() / s
            e_t = (g32[name].grad.double() - g64[name].grad).abs().max().item() / s
            flag = "  <<<" if e_h > 5 * e_t + 1e-6 else ""
            print(f"    {name:45s} hip {e_h:.2e}  torch32 {e_t:.2e}{flag}")
