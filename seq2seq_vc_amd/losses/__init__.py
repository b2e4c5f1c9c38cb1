"""Loss classes resolved by name from YAML (`criterions:`), as `seq2seq_vc.losses` is
(reference bin/vc_train.py:397-403).  Each forward is a fused HIP kernel (csrc/loss.hip, csrc/ctc.hip)."""
import torch

from ..modules import Lens
from ..ops import functional as Fn


class Seq2SeqLoss(torch.nn.Module):
    """masked L1(after)+L1(before) and BCE-with-logits(pos_weight) -- reference losses/seq2seq_loss.py:13-59."""

    def __init__(self, bce_pos_weight=10.0):
        super().__init__()
        self.bce_pos_weight = float(bce_pos_weight)

    def forward(self, after_outs, before_outs, logits, ys, labels, olens):
        ol = Lens.of(olens, ys.device)
        return Fn.seq2seq_loss(after_outs, before_outs, logits, ys, labels, ol.dev, self.bce_pos_weight)


class L1Loss(torch.nn.Module):
    """masked L1(before) + L1(after) -- reference losses/l1_loss.py:5-49."""

    def __init__(self, use_masking=True, reduction="mean"):
        super().__init__()
        if not use_masking or reduction != "mean":
            raise NotImplementedError("only use_masking=True, reduction='mean' (all recipes)")

    def forward(self, after_outs, before_outs, ys, olens):
        ol = Lens.of(olens, ys.device)
        l1, _ = Fn.seq2seq_loss(after_outs, before_outs, None, ys, None, ol.dev, 1.0)
        return l1


class GuidedMultiHeadAttentionLoss(torch.nn.Module):
    """reference losses/guided_attention_loss.py:130-165; att_ws (B, H, T_out, T_in)."""

    def __init__(self, sigma=0.4, alpha=1.0, reset_always=True):
        super().__init__()
        self.sigma, self.alpha = sigma, alpha

    def forward(self, att_ws, ilens, olens):
        dev = att_ws.device
        return Fn.guided_attention_loss(att_ws, Lens.of(ilens, dev).dev, Lens.of(olens, dev).dev, self.sigma, self.alpha)


class GuidedAttentionLoss(GuidedMultiHeadAttentionLoss):
    """reference losses/guided_attention_loss.py:6-127; att_ws (B, T_out, T_in)."""

    def forward(self, att_ws, ilens, olens):
        return super().forward(att_ws.unsqueeze(1), ilens, olens)


class StochasticDurationPredictorLoss(object):
    """Dummy, as in the reference (losses/duration_predictor_loss.py:59-63)."""

    def __call__(self, *args, **kwargs):
        return None


from ._aas import DurationPredictorLoss, ForwardSumLoss  # noqa: E402,F401
