"""Duration predictors of AAS-VC.

* `DurationPredictor` (reference modules/duration_predictor.py:27-128) runs on the HIP kernels.
* `StochasticDurationPredictor` (VITS flows; reference modules/duration_predictor.py:131-304,
  modules/vits/flow.py, modules/vits/transform.py) is ~200 launch-bound micro-ops on (B, 2..384, T_text<=64)
  tensors.  ROUND-1 STATUS: its arithmetic is expressed with stock torch GPU ops (fp32) -- the one part of
  the hot path not yet on hand-written kernels (SURVEY 2b K13 "leave on ATen initially"; DESIGN.md lists it
  as open).  The noise draw is injectable so parity with the reference is exact in distribution AND value.
"""
import math

import torch
import torch.nn.functional as TF
from torch import nn

from . import modules as Mo
from .ops import functional as Fn


class DurationPredictor(nn.Module):
    """(Conv1d k -> ReLU -> LayerNorm(channels) -> dropout) x n -> Linear(1)."""

    def __init__(self, idim, n_layers=2, n_chans=384, kernel_size=3, dropout_rate=0.1, offset=1.0):
        super().__init__()
        self.offset = offset
        self.conv = nn.ModuleList()
        for idx in range(n_layers):
            in_chans = idim if idx == 0 else n_chans
            self.conv += [nn.Sequential(nn.Conv1d(in_chans, n_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2),
                                        nn.ReLU(), Mo.LayerNorm(n_chans, dim=1), nn.Dropout(dropout_rate))]
        self.linear = nn.Linear(n_chans, 1)
        self.dropout_rate = dropout_rate

    def _forward(self, xs, x_lens=None, is_inference=False):
        p = self.dropout_rate if self.training else 0.0
        for blk in self.conv:
            xs = Fn.conv1d(xs, blk[0].weight, blk[0].bias, act="relu")
            xs = Fn.layer_norm(xs, blk[2].weight, blk[2].bias, blk[2].eps)   # LayerNorm over channels == last dim here
            xs = Fn.dropout(xs, p)
        out = Fn.linear(xs, self.linear.weight, self.linear.bias).squeeze(-1).float()   # (B, T) log-domain
        if is_inference:
            out = torch.clamp(torch.round(out.exp() - self.offset), min=0).long()
        if x_lens is not None:
            T = out.shape[1]
            mask = torch.arange(T, device=out.device)[None, :] < x_lens.dev[:, None]
            out = out * mask
        return out

    def forward(self, xs, x_lens=None):
        return self._forward(xs, x_lens, False)

    def inference(self, xs, x_lens=None):
        return self._forward(xs, x_lens, True)


# ---------------------------------------------------------------------------------------------------------
# VITS flow pieces (state_dict-compatible holders; forward in torch ops, fp32)
# ---------------------------------------------------------------------------------------------------------
class _Transpose(nn.Module):
    def __init__(self, d1, d2):
        super().__init__()
        self.d1, self.d2 = d1, d2

    def forward(self, x):
        return x.transpose(self.d1, self.d2)


class DilatedDepthSeparableConv(nn.Module):
    def __init__(self, channels, kernel_size, layers, dropout_rate=0.0, eps=1e-5):
        super().__init__()
        self.convs = nn.ModuleList()
        for i in range(layers):
            dilation = kernel_size ** i
            padding = (kernel_size * dilation - dilation) // 2
            self.convs += [nn.Sequential(
                nn.Conv1d(channels, channels, kernel_size, groups=channels, dilation=dilation, padding=padding),
                _Transpose(1, 2), nn.LayerNorm(channels, eps=eps, elementwise_affine=True), _Transpose(1, 2), nn.GELU(),
                nn.Conv1d(channels, channels, 1),
                _Transpose(1, 2), nn.LayerNorm(channels, eps=eps, elementwise_affine=True), _Transpose(1, 2), nn.GELU(),
                nn.Dropout(dropout_rate))]

    def forward(self, x, x_mask, g=None):
        if g is not None:
            x = x + g
        for f in self.convs:
            x = x + f(x * x_mask)
        return x * x_mask


def _rq_spline(x, uw, uh, ud, inverse, bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    nb = uw.shape[-1]
    inside = (x >= -bound) & (x <= bound)
    const = math.log(math.exp(1 - min_d) - 1)
    ud = TF.pad(ud, (1, 1))
    ud[..., 0] = const
    ud[..., -1] = const
    xin = torch.where(inside, x, torch.zeros_like(x))

    def knots(u, mn):
        s = mn + (1 - mn * nb) * TF.softmax(u, dim=-1)
        cs = TF.pad(torch.cumsum(s, dim=-1), (1, 0), value=0.0)
        cs = 2 * bound * cs - bound
        cs[..., 0] = -bound
        cs[..., -1] = bound
        return cs, cs[..., 1:] - cs[..., :-1]

    cw, w = knots(uw, min_w)
    ch, h = knots(uh, min_h)
    d = min_d + TF.softplus(ud)
    loc = (ch if inverse else cw).detach().clone()
    loc[..., -1] += 1e-6
    idx = (torch.sum(xin[..., None] >= loc, dim=-1) - 1)[..., None]
    pick = lambda a: a.gather(-1, idx)[..., 0]
    in_cw, in_w, in_ch, in_h = pick(cw), pick(w), pick(ch), pick(h)
    delta = h / w
    in_delta, in_d, in_d1 = pick(delta), pick(d), pick(d[..., 1:])
    if inverse:
        a = (xin - in_ch) * (in_d + in_d1 - 2 * in_delta) + in_h * (in_delta - in_d)
        b = in_h * in_d - (xin - in_ch) * (in_d + in_d1 - 2 * in_delta)
        c = -in_delta * (xin - in_ch)
        root = (2 * c) / (-b - torch.sqrt(b.pow(2) - 4 * a * c))
        out = root * in_w + in_cw
        tt = root * (1 - root)
        den = in_delta + (in_d + in_d1 - 2 * in_delta) * tt
        num = in_delta.pow(2) * (in_d1 * root.pow(2) + 2 * in_delta * tt + in_d * (1 - root).pow(2))
        lad = -(torch.log(num) - 2 * torch.log(den))
    else:
        th = (xin - in_cw) / in_w
        tt = th * (1 - th)
        den = in_delta + (in_d + in_d1 - 2 * in_delta) * tt
        out = in_ch + in_h * (in_delta * th.pow(2) + in_d * tt) / den
        num = in_delta.pow(2) * (in_d1 * th.pow(2) + 2 * in_delta * tt + in_d * (1 - th).pow(2))
        lad = torch.log(num) - 2 * torch.log(den)
    return torch.where(inside, out, x), torch.where(inside, lad, torch.zeros_like(lad))


class ConvFlow(nn.Module):
    def __init__(self, in_channels, hidden_channels, kernel_size, layers, bins=10, tail_bound=5.0):
        super().__init__()
        self.half_channels = in_channels // 2
        self.hidden_channels, self.bins, self.tail_bound = hidden_channels, bins, tail_bound
        self.input_conv = nn.Conv1d(self.half_channels, hidden_channels, 1)
        self.dds_conv = DilatedDepthSeparableConv(hidden_channels, kernel_size, layers, dropout_rate=0.0)
        self.proj = nn.Conv1d(hidden_channels, self.half_channels * (bins * 3 - 1), 1)
        self.proj.weight.data.zero_()
        self.proj.bias.data.zero_()

    def forward(self, x, x_mask, g=None, inverse=False):
        xa, xb = x.split(x.size(1) // 2, 1)
        h = self.dds_conv(self.input_conv(xa), x_mask, g=g)
        h = self.proj(h) * x_mask
        b, c, t = xa.shape
        h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
        den = math.sqrt(self.hidden_channels)
        xb, lad = _rq_spline(xb, h[..., : self.bins] / den, h[..., self.bins: 2 * self.bins] / den, h[..., 2 * self.bins:],
                             inverse, self.tail_bound)
        x = torch.cat([xa, xb], 1) * x_mask
        return (x, torch.sum(lad * x_mask, [1, 2])) if not inverse else x


class ElementwiseAffineFlow(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.register_parameter("m", nn.Parameter(torch.zeros(channels, 1)))
        self.register_parameter("logs", nn.Parameter(torch.zeros(channels, 1)))

    def forward(self, x, x_mask, inverse=False, **kwargs):
        if not inverse:
            return (self.m + torch.exp(self.logs) * x) * x_mask, torch.sum(self.logs * x_mask, [1, 2])
        return (x - self.m) * torch.exp(-self.logs) * x_mask


class FlipFlow(nn.Module):
    def forward(self, x, *args, inverse=False, **kwargs):
        x = torch.flip(x, [1])
        return (x, x.new_zeros(x.size(0))) if not inverse else x


class LogFlow(nn.Module):
    def forward(self, x, x_mask, inverse=False, eps=1e-5, **kwargs):
        if not inverse:
            y = torch.log(torch.clamp_min(x, eps)) * x_mask
            return y, torch.sum(-y, [1, 2])
        return torch.exp(x) * x_mask


class StochasticDurationPredictor(nn.Module):
    def __init__(self, channels=192, kernel_size=3, dropout_rate=0.5, flows=4, dds_conv_layers=3, global_channels=-1):
        super().__init__()
        self.pre = nn.Conv1d(channels, channels, 1)
        self.dds = DilatedDepthSeparableConv(channels, kernel_size, layers=dds_conv_layers, dropout_rate=dropout_rate)
        self.proj = nn.Conv1d(channels, channels, 1)
        self.log_flow = LogFlow()
        self.flows = nn.ModuleList([ElementwiseAffineFlow(2)])
        for _ in range(flows):
            self.flows += [ConvFlow(2, channels, kernel_size, layers=dds_conv_layers), FlipFlow()]
        self.post_pre = nn.Conv1d(1, channels, 1)
        self.post_dds = DilatedDepthSeparableConv(channels, kernel_size, layers=dds_conv_layers, dropout_rate=dropout_rate)
        self.post_proj = nn.Conv1d(channels, channels, 1)
        self.post_flows = nn.ModuleList([ElementwiseAffineFlow(2)])
        for _ in range(flows):
            self.post_flows += [ConvFlow(2, channels, kernel_size, layers=dds_conv_layers), FlipFlow()]
        if global_channels > 0:
            self.global_conv = nn.Conv1d(global_channels, channels, 1)
        self.noise = None   # set to a (B, 2, T) tensor to inject the draw (parity tests); consumed once

    def _randn(self, shape, device):
        if self.noise is not None:
            n, self.noise = self.noise.to(device=device, dtype=torch.float32), None
            return n
        return torch.randn(shape, device=device, dtype=torch.float32)

    def forward(self, x, x_mask, w=None, g=None, inverse=False, noise_scale=1.0):
        """x (B, C, T) (detached: no gradient to the encoder), x_mask (B,1,T), w (B,1,T) -> NLL (B,) or durations."""
        x = x.detach().float()
        x_mask = x_mask.float()
        x = self.pre(x)
        if g is not None:
            x = x + self.global_conv(g.detach())
        x = self.dds(x, x_mask)
        x = self.proj(x) * x_mask
        if not inverse:
            assert w is not None, "w must be provided."
            w = w.float()
            h_w = self.post_proj(self.post_dds(self.post_pre(w), x_mask)) * x_mask
            e_q = self._randn((w.size(0), 2, w.size(2)), x.device) * x_mask
            z_q, logdet_tot_q = e_q, 0.0
            for flow in self.post_flows:
                z_q, logdet_q = flow(z_q, x_mask, g=(x + h_w))
                logdet_tot_q = logdet_tot_q + logdet_q
            z_u, z1 = torch.split(z_q, [1, 1], 1)
            u = torch.sigmoid(z_u) * x_mask
            z0 = (w - u) * x_mask
            logdet_tot_q = logdet_tot_q + torch.sum((TF.logsigmoid(z_u) + TF.logsigmoid(-z_u)) * x_mask, [1, 2])
            logq = torch.sum(-0.5 * (math.log(2 * math.pi) + (e_q ** 2)) * x_mask, [1, 2]) - logdet_tot_q
            z0, logdet_tot = self.log_flow(z0, x_mask)
            z = torch.cat([z0, z1], 1)
            for flow in self.flows:
                z, logdet = flow(z, x_mask, g=x, inverse=inverse)
                logdet_tot = logdet_tot + logdet
            nll = torch.sum(0.5 * (math.log(2 * math.pi) + (z ** 2)) * x_mask, [1, 2]) - logdet_tot
            return nll + logq
        flows = list(reversed(self.flows))
        flows = flows[:-2] + [flows[-1]]
        z = self._randn((x.size(0), 2, x.size(2)), x.device) * noise_scale
        for flow in flows:
            z = flow(z, x_mask, g=x, inverse=inverse)
        z0, _ = z.split(1, 1)
        return torch.ceil(torch.exp(z0) * x_mask)
