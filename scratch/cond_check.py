"""How well conditioned are the reference's own gradients?  fp32 torch CPU (the golden) vs the same graph in fp64."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from conftest import load_golden
from make_slices import GRAD_SLICES
from oracle import tta as otta, unet as ounet
size = 32
g = load_golden(f"full_{size}")
torch.set_num_threads(8)
om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(), 7), 8).double()
sel = torch.arange(16) * 3
torch.manual_seed(int(g["img_seed"]))
imgs = torch.randn(1, 1, size, size, size)
outs = {}
for br in ("a", "b"):
    torch.manual_seed(int(g[f"seed_{br}"]))
    d = otta.draw_branch(1, [size] * 3)
    import torch.nn.functional as F
    from oracle import gin as ogin, mind as omind
    x = ogin.gin_chain(imgs, *d["gin_draw"])
    r, rinv = otta.rand_affine_from_draw(d["affine_draw"])
    x = omind.mind3d(otta.warp(x, r, "border"), d["mind_noise"])
    y = om(x.double())[:, sel]
    size_ = [1, 1, size, size, size]
    ident = F.affine_grid(torch.eye(4)[:3][None], size_, align_corners=False)
    grid = (0.0 * ident + (F.affine_grid(rinv, size_, align_corners=False) - ident)) + ident
    outs[br] = F.grid_sample(y, grid.double(), padding_mode="zeros", align_corners=False)
    st = int(g["slice_step"])
    print(br, "logits fp32 vs fp64:", (outs[br][:, :, ::st, ::st, ::st].float() - g[f"out_{br}_slice"]).abs().max().item())
loss = otta.consistency_loss(outs["a"], outs["b"])
print("loss", loss.item(), float(g["loss"]))
loss.backward()
for key in [k for k in g if k.startswith("g::")]:
    name = key[3:]
    got = dict(om.named_parameters())[name].grad
    if name in GRAD_SLICES:
        got = got[GRAD_SLICES[name]]
    scale = float(g[f"gmax::{name}"])
    rel = (got.float() - g[key]).abs().max().item() / scale
    if "norm" not in name or rel > 1e-3:
        print(f"  {name:55s} fp32-ref vs fp64 rel {rel:.3e}")
