"""Follow-up of step_loss_ab.py: the network inputs of ONE loss call of the referee workload (captured in the fp32 run) through
the fp32 and the 16-bit product nets, and through the CPU oracle with every block's output rounded to the 16-bit type
(emulated storage): per-sample logit error, and the first block whose rounded output moves the logits.
usage: step_forward_ab.py [fp16|bf16] [call index]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import dg_tta_amd.tta.tta as T
from dg_tta_amd.optim import HipAdamW
from dg_tta_amd.tta.model_utils import get_model_from_network
from dg_tta_amd.tta.torch_utils import release_resident
from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
from dg_tta_amd.utils import disable_internal_augmentation
from oracle.replay import cpu_rng_for_device_draws

other = sys.argv[1] if len(sys.argv) > 1 else "fp16"
CALL = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
args = bench.parse_args(["--referee-epochs", "2", "--referee-accum", "16"])
args.size = args.referee_patch
args.pretrain_hooks, args.target_noise, args.lr = "GIN_MIND", None, 1e-5
state, prep = bench.pretrained_weights(args, dev)
P, A, seed = [args.referee_patch] * 3, 16, 6006
captured = {}
orig_run = T.run_both_branches


def build(dtype):
    net, cfg, mapping, modmod, data = bench.build_workload(args, dev, 0, dtype)
    cfg.update(epochs=2, patches_to_be_accumulated=A)
    names = cfg["optimized_labels"]
    model = get_model_from_network(net, modmod, None)
    fused = _fuse_head_if_possible(model, modmod, mapping, names)
    return model, cfg, mapping, modmod, data, fused, names


model32, cfg, mapping, modmod, data, fused, names = build("fp32")
count = [0]


def spy(prepared, *a, **k):
    if count[0] == CALL:
        captured["prepared"] = {kk: (v.clone() if torch.is_tensor(v) else ([t.clone() if torch.is_tensor(t) else t for t in v] if isinstance(v, (list, tuple)) else v))
                                for kk, v in prepared.items()}
    count[0] += 1
    return orig_run(prepared, *a, **k)


T.run_both_branches = spy
opt = HipAdamW(model32.parameters(), lr=cfg["lr"], grad_scale=model32.loss_scale)
disable_internal_augmentation()
release_resident()
with cpu_rng_for_device_draws():
    torch.manual_seed(seed)
    np.random.seed(seed)
    tta_unit(model32, opt, cfg, [data], P, mapping, modmod, dev, fused)
T.run_both_branches = orig_run
release_resident()
pr = captured["prepared"]
print("captured call", CALL, {k: (tuple(v.shape), str(v.dtype)) if torch.is_tensor(v) else type(v).__name__ for k, v in pr.items()})
x = pr["x"]
print("network input x: min %.4g max %.4g mean %.4g; per sample max |x|:" % (float(x.min()), float(x.max()), float(x.mean())),
      [round(float(v), 3) for v in x.abs().flatten(1).max(1).values])

# the weights BEFORE the optimizer step of that epoch = the pre-trained ones at lr 1e-5 to ~1e-5: rebuild both nets fresh
model32b, *_ = build("fp32")
model16, *_ = build(other)


def fwd(model, prepared):
    p = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in prepared.items()}
    if p.get("feat") is not None:
        adt = next(iter(model.parameters())).dtype
        p["feat"] = p["feat"].to(getattr(model, "act_dtype", None) or getattr(getattr(model, "module", model), "act_dtype", torch.float32))
    with torch.no_grad():
        model.train()
        ta, tb = orig_run(p, cfg, model, mapping, names, modmod, fused, steps=4)
    return torch.cat([ta, tb], 0).float().cpu()


l32 = fwd(model32b, pr)
l16 = fwd(model16, pr)
rng = float(l32.abs().max())
print("warped logits: range %.3f; per-sample max |delta| / range:" % rng, [f"{float(v) / rng:.2e}" for v in (l32 - l16).abs().flatten(1).max(1).values])
print("per-sample rms delta / range:", [f"{float(v) / rng:.2e}" for v in (l32 - l16).pow(2).flatten(1).mean(1).sqrt()])
for st in range(4):
    m32 = (l32[st].sum(0) > 0) & (l32[st + 4].sum(0) > 0)
    m16 = (l16[st].sum(0) > 0) & (l16[st + 4].sum(0) > 0)
    s32a, s32b = l32[st].sum(0), l32[st + 4].sum(0)
    near = ((s32a.abs() < 2e-3 * rng) | (s32b.abs() < 2e-3 * rng)).sum()
    print(f"step {st}: consistency mask fp32 {int(m32.sum())} voxels of {m32.numel()}, {other} {int(m16.sum())}, differing {int((m32 != m16).sum())}; "
          f"voxels whose mapped-logit sum is within 2e-3 of the range of 0 in a branch: {int(near)}")

# the same call captured in the 16-bit run: are the network INPUTS the same?
count[0] = 0
captured.clear()
T.run_both_branches = spy
opt16 = HipAdamW(model16.parameters(), lr=cfg["lr"], grad_scale=model16.loss_scale)
disable_internal_augmentation()
release_resident()
with cpu_rng_for_device_draws():
    torch.manual_seed(seed)
    np.random.seed(seed)
    tta_unit(model16, opt16, cfg, [data], P, mapping, modmod, dev, fused)
T.run_both_branches = orig_run
pr16 = captured["prepared"]
print(f"{other} run, same call: x equal {bool(torch.equal(pr16['x'], pr['x']))}, max |x delta| {float((pr16['x'] - pr['x']).abs().max()):.3e}; inverse maps equal "
      f"{all(bool(torch.equal(a, b)) for a, b in zip(pr16['inverses'], pr['inverses']))}")
f16, f32 = pr16["feat"].float(), pr["feat"].float()
print("MIND features: per-sample max |delta|", [f"{float(v):.3e}" for v in (f16 - f32).abs().flatten(1).max(1).values],
      "per-sample mean |delta|", [f"{float(v):.3e}" for v in (f16 - f32).abs().flatten(1).mean(1)])

# the captured inputs once more, this time the way the epoch runs them: training mode WITH the autograd graph


def fwd_grad(model, prepared):
    p = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in prepared.items()}
    model.train()
    ta, tb = orig_run(p, cfg, model, mapping, names, modmod, fused, steps=4)
    from dg_tta_amd import ops as O
    loss, dice = O.consistency_loss(ta, tb, 1)
    return torch.cat([ta, tb], 0).detach().float().cpu(), dice.detach().float().cpu().reshape(4, -1)


from dg_tta_amd import _lib
lib = _lib.load()
g32, d32 = fwd_grad(model32b, pr)
model16, *_ = build(other)          # (fresh: the one above has taken the epoch's optimizer step)
n16 = fwd(model16, pr16)
print("fresh 16-bit net, no graph: per-sample max |logit delta| / range", [f"{float(v) / rng:.2e}" for v in (l32 - n16).abs().flatten(1).max(1).values])
for env in [None] + [e for e in os.environ.get("AB_SWITCHES", "").split(",") if e]:
    if env:
        k, v = env.split("=")
        os.environ[k] = v
        lib.dgtta_reload_env()
    g16, d16 = fwd_grad(model16, pr16)
    print(f"with the graph ({env or 'defaults'}): per-sample max |logit delta| / range", [f"{float(v) / rng:.2e}" for v in (g32 - g16).abs().flatten(1).max(1).values],
          "| per-step max |dice delta|", [f"{float(v):.2e}" for v in (d32 - d16).abs().max(1).values])
    if env:
        os.environ.pop(k)
        lib.dgtta_reload_env()

# the loss of both logit sets evaluated by torch on the host (oracle/tta.py:consistency_loss per step)
from oracle import tta as otta


def host_dice(l):
    out = []
    for st in range(4):
        ta, tb = l[st:st + 1], l[st + 4:st + 5]
        mask = (ta.sum(1, keepdim=True) > 0.0).float() * (tb.sum(1, keepdim=True) > 0.0).float()
        out.append(otta.soft_dice_loss(ta.softmax(1) * mask, tb.softmax(1) * mask)[0])
    return torch.stack(out)


h32, h16 = host_dice(g32), host_dice(g16)
print("host evaluation of the product's logits: per-step max |dice(fp32 logits) - dice(16-bit logits)|", [f"{float(v):.2e}" for v in (h32 - h16).abs().max(1).values])
print("kernel dice vs host dice on the same logits: fp32", [f"{float(v):.2e}" for v in (h32 - d32).abs().max(1).values], other, [f"{float(v):.2e}" for v in (h16 - d16).abs().max(1).values])
print("step 3, classes 1..: host fp32", [round(float(v), 4) for v in h32[3][1:8]], "host 16-bit", [round(float(v), 4) for v in h16[3][1:8]], "kernel 16-bit", [round(float(v), 4) for v in d16[3][1:8]])
