"""AdamW on the multi-tensor HIP kernel (csrc/adamw.hip).  Stand-in for `torch.optim.AdamW(model.parameters(), lr)`
of dg_tta/tta/tta.py:185 with PyTorch's defaults; state layout (`step`, `exp_avg`, `exp_avg_sq`) matches torch's so
optimizer state dicts are interchangeable.  Parameters whose .grad is None are skipped, as PyTorch does."""
import torch

from . import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        """grad_scale: static loss scale carried by the gradients (fp16 storage path, HipPlainConvUNet.loss_scale); the
        kernel divides it out on load, so state and parameters are those of the unscaled problem."""
        self.grad_scale = float(grad_scale)
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("HipAdamW does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("HipAdamW needs contiguous parameters and gradients")
                by_step.setdefault(st["step"], []).append(p)
            for step, ps in by_step.items():
                ops.adamw_step([p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                               [self.state[p]["exp_avg_sq"] for p in ps], step, group["lr"], group["betas"],
                               group["eps"], group["weight_decay"], self.grad_scale)
                # the kernel wrote through raw pointers: tell autograd / weight caches that the tensors changed
                torch.autograd.graph.increment_version(ps)
        return loss
