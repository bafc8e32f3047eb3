import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from conftest import load_golden, unpack_draws, state_from_golden
import test_gpu_tta as T
from dg_tta_amd import ops
from dg_tta_amd.mind import MIND3D
from dg_tta_amd.optim import HipAdamW
from dg_tta_amd.tta.torch_utils import fix_all, release_all
g = load_golden("tta_epoch")
model = T._model(g); model.set_selected_classes(g["map_idxs"])
opt = HipAdamW(model.parameters(), lr=float(g["lr"]))
imgs = g["imgs"].to("cuda:0"); inv = torch.full((), 0.5, device="cuda:0")
model.apply(fix_all)
for epoch in range(3):
    if epoch == 1: model.apply(release_all)
    for acc in range(2):
        ta = T.hip_branch(model, imgs, unpack_draws(g, f"e{epoch}s{acc}_a")); tb = T.hip_branch(model, imgs, unpack_draws(g, f"e{epoch}s{acc}_b"))
        loss,_ = ops.consistency_loss(ta, tb, 1); print(epoch, acc, float(loss), float(g["losses"][epoch*2+acc]))
        if epoch>=1: torch.autograd.backward(loss, grad_tensors=inv)
    if epoch>=1:
        if epoch==1:
            # compare accumulated grads sign agreement with post-pre direction
            post = state_from_golden(g,"p::"); pre = state_from_golden(g,"w::")
        opt.step(); opt.zero_grad()
with torch.no_grad():
    logits = model(MIND3D()(imgs, g["eval_noise"].to("cuda:0"))).cpu()
ref = g["eval_logits"]
d=(logits-ref).abs()
print("logit max err", d.max().item(), "mean", d.mean().item(), "ref absmax", ref.abs().max().item())
top2 = ref.topk(2, dim=1).values; margin=(top2[:,0]-top2[:,1])
mis = logits.argmax(1)!=ref.argmax(1)
print("mismatch frac", mis.float().mean().item(), "max margin at mismatches", margin[mis].max().item() if mis.any() else 0)
post = state_from_golden(g,"p::"); pre = state_from_golden(g,"w::")
for name,p in model.state_dict().items():
    if name in post and ".all_modules." not in name and not name.startswith("decoder.encoder"):
        dref = post[name]-pre[name]; dd = p.cpu()-pre[name]
        if dref.abs().max()>0:
            print(f"{name:50s} ref|d| {dref.abs().mean():.2e} ours|d| {dd.abs().mean():.2e} diff {(dd-dref).abs().mean():.2e} signagree {((dd*dref)>0).float().mean():.3f}")
