"""Hot-path utilities with the reference's names (reference seq2seq_vc/utils/duration_calculator.py)."""
import torch

from ..ops import kernels as K
from ..ops import kernels_aas as KA


class DurationCalculator(torch.nn.Module):
    """Durations of a teacher model's attention (utils/duration_calculator.py:13-65): `att_ws` (T_feats, T_text) or
    (#layers, #heads, T_feats, T_text) -> (durations LongTensor (T_text,), focus rate).  One HIP launch (csrc/lenreg.hip):
    most diagonal head by mean row maximum, then the per-input count of arg-max frames."""

    @torch.no_grad()
    def forward(self, att_ws):
        if att_ws.dim() == 2:
            a = att_ws[None]
        elif att_ws.dim() == 4:
            a = att_ws.reshape(-1, att_ws.shape[-2], att_ws.shape[-1])
        else:
            raise ValueError("att_ws should be 2 or 4 dimensional tensor.")
        if not a.is_cuda:
            raise RuntimeError("DurationCalculator runs on the GPU (there is no CPU path)")
        a = K.cast(a.contiguous(), torch.float32)
        dur, focus, _ = KA.attn_durations(a)
        return dur, focus
