"""AAS-VC losses: forward-sum (CTC over the attention matrix) and the duration-predictor MSE."""
import math

import torch

from ..modules import Lens
from ..ops import functional_aas as FA
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
        return FA.forward_sum_loss(log_p_attn, self._prior(il, ol, Tf, Tx, dev), il.dev, ol.dev, blank_prob)


class DurationPredictorLoss(torch.nn.Module):
    """MSE in the log domain over non-padded tokens (reference losses/duration_predictor_loss.py:5-57).
    (B, T_text) scalars per step: evaluated with elementwise torch ops on the device."""

    def __init__(self, use_masking=True, offset=1.0, reduction="mean"):
        super().__init__()
        self.offset, self.use_masking, self.reduction = offset, use_masking, reduction

    def forward(self, d_outs, ds, ilens):
        il = Lens.of(ilens, ds.device)
        tgt = torch.log(ds.float() + self.offset)
        err = (d_outs.float() - tgt) ** 2
        if self.use_masking:
            mask = torch.arange(ds.shape[1], device=ds.device)[None, :] < il.dev[:, None]
            n = mask.sum()
            return (err * mask).sum() / n if self.reduction == "mean" else (err * mask).sum()
        return err.mean() if self.reduction == "mean" else err.sum()
