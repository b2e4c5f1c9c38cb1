"""Flat-buffer Adam with gradient clipping and WarmupLR, one fused HIP launch sequence per step.

Replaces the reference's `clip_grad_norm_` -> `torch.optim.Adam.step()` -> `WarmupLR.step()` sequence
(trainers/ar_vc.py:99-107, trainers/aas_vc.py:151-158, schedulers/warmup_lr.py:54-61).

All trainable parameters of the model are re-pointed into ONE contiguous fp32 buffer (`flat_p`); their
gradients live in a parallel buffer (`flat_g`, exposed as `p.grad` views and as `p._s2s_grad` so the
wgrad kernels accumulate straight into it); Adam moments and the bf16 shadow used by bf16 GEMMs are
flat as well.  The step counter, learning rate, gradient norm and clip coefficient live in a 4-float
device tensor, so the optimiser step is hipGraph-capturable and needs no host synchronisation.  The
flat gradient buffer is also what data-parallel training all-reduces (distributed.py): a handful of
large RCCL collectives instead of one per tensor.
"""
import torch

from .ops import kernels as K


class FlatAdam:
    def __init__(self, model, lr=8e-5, betas=(0.9, 0.999), eps=1e-8, grad_norm=1.0, warmup_steps=4000, bf16_shadow=False,
                 align=64):
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs the model on the GPU (there is no CPU path)")
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.grad_norm, self.warmup_steps = float(grad_norm), float(warmup_steps or 0)
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.offsets, self.numel = offs, n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)  # step, lr, grad_norm, clip_coef
        self.partial = torch.empty(1024, dtype=torch.float64, device=dev)
        for p, o in zip(self.params, offs):
            k = p.numel()
            self.flat_p[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p._s2s_grad = self.flat_g[o:o + k].view(p.shape)
            p.grad = p._s2s_grad
            if self.shadow is not None:
                p._s2s_bf16 = self.shadow[o:o + k].view(p.shape)
        if self.shadow is not None:
            self.refresh_shadow()

    def refresh_shadow(self):
        """bf16 copy of the fp32 master weights (after loading a checkpoint / at start)."""
        if self.shadow is not None:
            self.shadow.copy_(K.cast(self.flat_p, torch.bfloat16))

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def step(self):
        K.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.shadow, self.state, self.partial, self.lr,
                    self.betas, self.eps, self.grad_norm, self.warmup_steps)

    # -- introspection (host sync; for logging / tests only) -------------------------------------
    def last_stats(self):
        s = self.state.tolist()
        return {"step": int(s[0]), "lr": s[1], "grad_norm": s[2], "clip_coef": s[3]}

    def state_dict(self):
        return {"step": self.state.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "offsets": list(self.offsets), "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "grad_norm": self.grad_norm, "warmup_steps": self.warmup_steps}

    def load_state_dict(self, sd):
        self.state.copy_(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.refresh_shadow()
