"""debug: per-layer gradient error of the full net vs golden for conv_impl 0/1."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from conftest import load_golden
from make_slices import GRAD_SLICES
import test_gpu_full_topology as T
from dg_tta_amd import ops
from oracle import tta as otta
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = load_golden(f"full_{size}")
for impl, dt in ((0, torch.float32), (1, torch.float32), (0, torch.bfloat16)):
    _, hm = T._nets(int(g["w_seed"]), dt, conv_impl=impl)
    hm.set_selected_classes(torch.arange(16) * 3)
    torch.manual_seed(int(g["img_seed"]))
    imgs = torch.randn(1, 1, size, size, size).to("cuda:0")
    outs = {}
    for br in ("a", "b"):
        torch.manual_seed(int(g[f"seed_{br}"]))
        outs[br] = T._hip_branch(hm, imgs, otta.draw_branch(1, [size] * 3))
        st = int(g["slice_step"])
        err = (outs[br].detach().cpu()[:, :, ::st, ::st, ::st] - g[f"out_{br}_slice"]).abs().max().item()
        print(f"impl {impl} {dt} branch {br}: logits err {err:.3e} range {float(g[f'out_{br}_absmax']):.2f}")
    loss, dice = ops.consistency_loss(outs["a"], outs["b"], 1)
    print(f"  loss {float(loss):.7f} ref {float(g['loss']):.7f}")
    loss.backward()
    named = dict(hm.named_parameters())
    for key in [k for k in g if k.startswith("g::")]:
        name = key[3:]
        got = named[name].grad.detach().cpu()
        if name in GRAD_SLICES:
            got = got[GRAD_SLICES[name]]
        scale = float(g[f"gmax::{name}"])
        rel = (got - g[key]).abs().max().item() / (scale + 1e-30)
        g64 = g[f"g64::{name}"]
        rel64 = (got.double() - g64).abs().max().item() / (scale + 1e-30)
        cos = torch.nn.functional.cosine_similarity(got.double().flatten(), g64.flatten(), dim=0).item()
        cosr = torch.nn.functional.cosine_similarity(g[key].double().flatten(), g64.flatten(), dim=0).item()
        sign = (torch.sign(got.double()) == torch.sign(g64)).float().mean().item()
        print(f"    {name:50s} vs ref32 {rel:.2e} vs f64 {rel64:.2e} (ref32's own {float(g['gcond::' + name]):.2e})  cos {cos:.5f} (ref32 {cosr:.5f}) sign {sign:.4f}")
