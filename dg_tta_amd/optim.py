"""AdamW on the multi-tensor HIP kernel (csrc/adamw.hip).  Stand-in for `torch.optim.AdamW(model.parameters(), lr)`
of dg_tta/tta/tta.py:185 with PyTorch's defaults; state layout (`step`, `exp_avg`, `exp_avg_sq`) matches torch's so
optimizer state dicts are interchangeable.  Parameters whose .grad is None are skipped, as PyTorch does."""
import torch

from . import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0,
                 min_grad_scale=1.0):
        """grad_scale: loss scale carried by the gradients (fp16 storage path, HipPlainConvUNet.loss_scale); the kernel
        divides it out on load, so state and parameters are those of the unscaled problem.  With grad_scale != 1 every
        step is guarded: when a gradient is inf / NaN the kernel leaves parameters and state untouched, and the next
        `resolve_overflow()` (called by tta_epoch at the start of an epoch, when the stream is drained anyway) halves the
        scale and takes the skipped step out of the step counters.  The guard belongs to the 16-bit path, not to the value
        of the scale: it stays on when the scale has decayed to min_grad_scale = 1."""
        self.grad_scale = float(grad_scale)
        self.guarded = float(grad_scale) != 1.0
        self.min_grad_scale = float(min_grad_scale)
        self.skipped_steps = 0
        self._flag = None               # int32 device scalar of the last guarded step
        self._pending = []              # parameters whose step counter that step advanced
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def resolve_overflow(self):
        """Reads the overflow flag of the last guarded step (one small device->host copy).  True: that step was skipped;
        the loss scale is halved (not below min_grad_scale) and the step counters are rolled back."""
        if self._flag is None:
            return False
        bad = bool(self._flag.item())
        if bad:
            for p in self._pending:
                self.state[p]["step"] = int(self.state[p]["step"]) - 1
            self.skipped_steps += 1
            self.grad_scale = max(self.grad_scale * 0.5, self.min_grad_scale)
        self._flag, self._pending = None, []
        return bad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # the gradients of THIS step carry the scale that was in force when they were produced: settling an unread flag of
        # the previous step may halve self.grad_scale, which then applies to the NEXT backward (a caller that reads
        # `grad_scale` for its loss only after step(), e.g. tta_epoch, sees one consistent value)
        scale_now = self.grad_scale
        self.resolve_overflow()
        guarded = self.guarded
        work = []                        # (group, step, params)
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
            work.extend((group, step, ps) for step, ps in by_step.items())
        flag = None
        if guarded and work:             # all or nothing: one flag over every gradient of this step
            flag = torch.zeros((), dtype=torch.int32, device=work[0][2][0].device)
            ops.grads_nonfinite([p.grad for _, _, ps in work for p in ps], flag)
        for group, step, ps in work:
            ops.adamw_step([p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                           [self.state[p]["exp_avg_sq"] for p in ps], step, group["lr"], group["betas"],
                           group["eps"], group["weight_decay"], scale_now, skip_flag=flag)
            # the kernel wrote through raw pointers: tell autograd / weight caches that the tensors changed
            torch.autograd.graph.increment_version(ps)
            if guarded:
                self._pending.extend(ps)
        self._flag = flag
        return loss
