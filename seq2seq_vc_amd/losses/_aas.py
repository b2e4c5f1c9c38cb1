"""AAS-VC losses: forward-sum (CTC over the attention matrix) and the duration-predictor MSE."""
import math
import os

import torch

from ..modules import Lens
from ..ops import functional_aas as FA
from ..ops import kernels as K
from ..ops import kernels_aas as KA


class ForwardSumLoss(torch.nn.Module):
    """reference losses/forward_sum_loss.py:12-116.  The beta-binomial log-prior is evaluated on the
    device (fp64 lgamma) and cached per (text_lens, feat_lens) like the reference's `_cache`; the
    per-utterance F.ctc_loss loop is one batched kernel."""

    def __init__(self, cache_prior: bool = True):
        super().__init__()
        self.cache_prior = cache_prior
        self._cache = {}

    def _prior(self, il, ol, Tf, Tx, device):
        if il.cap is not None or ol.cap is not None:   # lengths are data of a captured step: evaluate inside the graph
            return KA.betabinom_prior(len(il.host), Tf, Tx, il.dev, ol.dev, device)
        key = (il.host, ol.host, Tf, Tx, str(device))
        if self.cache_prior and key in self._cache:
            return self._cache[key]
        prior = KA.betabinom_prior(len(il.host), Tf, Tx, il.dev, ol.dev, device)
        if self.cache_prior:
            if len(self._cache) > 256:
                self._cache.clear()
            self._cache[key] = prior
        return prior

    def forward(self, log_p_attn, ilens, olens, blank_prob: float = math.e ** -1):
        dev = log_p_attn.device
        il, ol = Lens.of(ilens, dev), Lens.of(olens, dev)
        _, Tf, Tx = log_p_attn.shape
        pre = getattr(log_p_attn, "_s2s_fs", None)
        if pre is not None and pre[2:] == (il.dev.data_ptr(), ol.dev.data_ptr(), blank_prob):
            return FA.forward_sum_loss_prefetched(log_p_attn, pre[0], pre[1])
        return FA.forward_sum_loss(log_p_attn, self._prior(il, ol, Tf, Tx, dev), il.dev, ol.dev, blank_prob)

    def prefetch(self, log_p_attn, ilens, olens, blank_prob: float = math.e ** -1):
        """Hook for the model (AASVC.forward_sum_prefetch, set by the trainers): called in the training forward pass as soon as
        the alignment module has produced `log_p_attn`.  The alpha recursion is T_feats dependent steps on one wavefront per
        utterance (~150 us that leave the chip empty); here it runs on the auxiliary stream (ops.functional.branch_run) beside
        the length regulator / decoder instead of after them, and forward() picks the result up from the tensor when it is
        called with the same lengths.  The model's branch_join covers it."""
        dev = log_p_attn.device
        if dev.type != "cuda" or os.environ.get("S2SVC_FS_PREFETCH", "1") == "0":       # (A/B aid)
            return
        from ..ops import functional as Fn
        il, ol = Lens.of(ilens, dev), Lens.of(olens, dev)
        _, Tf, Tx = log_p_attn.shape
        lp = log_p_attn.detach()

        def run():
            return KA.forward_sum(lp.float().contiguous(), self._prior(il, ol, Tf, Tx, dev), il.dev, ol.dev, blank_prob)

        loss_b, grad = Fn.branch_run(run, uses=(lp, il.dev, ol.dev))
        log_p_attn._s2s_fs = (loss_b, grad, il.dev.data_ptr(), ol.dev.data_ptr(), blank_prob)


class _DurLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d_outs, ds, lens, offset, mean):
        from .. import _lib
        d, t = d_outs.float().contiguous(), ds.float().contiguous()
        B, T = d.shape
        stats = torch.empty(1, dtype=torch.float32, device=d.device)
        out = torch.empty((), dtype=torch.float32, device=d.device)
        _lib.check(_lib.lib().s2svc_duration_loss_fwd(B, T, K.ptr(d), K.ptr(t), K.ptr(lens), offset, 1 if mean else 0, K.ptr(stats),
                                                      K.ptr(out), K.stream()), "duration_loss_fwd")
        ctx.meta = (lens, offset, mean, d_outs.dtype)
        ctx.save_for_backward(d, t, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        d, t, stats = ctx.saved_tensors
        lens, offset, mean, dtype = ctx.meta
        B, T = d.shape
        dd = torch.empty_like(d)
        _lib.check(_lib.lib().s2svc_duration_loss_bwd(B, T, K.ptr(d), K.ptr(t), K.ptr(lens), offset, 1 if mean else 0, K.ptr(stats),
                                                      K.ptr(g.float().contiguous()), K.ptr(dd), K.stream()), "duration_loss_bwd")
        return (dd if dtype == torch.float32 else K.cast(dd, dtype)), None, None, None, None


class DurationPredictorLoss(torch.nn.Module):
    """MSE in the log domain over non-padded tokens (reference losses/duration_predictor_loss.py:5-57), one fused
    kernel each way (csrc/loss_dur.hip)."""

    def __init__(self, use_masking=True, offset=1.0, reduction="mean"):
        super().__init__()
        if reduction not in ("mean", "sum"):
            raise NotImplementedError("reduction must be 'mean' or 'sum'")
        self.offset, self.use_masking, self.reduction = offset, use_masking, reduction

    def forward(self, d_outs, ds, ilens):
        lens = Lens.of(ilens, ds.device).dev if self.use_masking else None
        return _DurLoss.apply(d_outs, ds, lens, float(self.offset), self.reduction == "mean")
