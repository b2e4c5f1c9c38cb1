"""Duration predictors of AAS-VC.

* `DurationPredictor` (reference modules/duration_predictor.py:27-128) runs on the HIP kernels.
* `StochasticDurationPredictor` (VITS flows; reference modules/duration_predictor.py:131-304,
  modules/vits/flow.py, modules/vits/transform.py) runs on csrc/sdp.hip (fused LayerNorm+GELU(+dropout+residual+
  mask), rank-1 expansion, rational-quadratic spline forward / inverse / backward, fused NLL glue) plus the shared
  GEMM and depthwise-conv kernels, in fp32.  The noise draw (torch.randn on the device, as the reference) is
  injectable so parity with the reference is exact in distribution AND value.
"""
import math

import torch
from torch import nn

from . import modules as Mo
from .ops import functional as Fn
from .ops import functional_aas as FA
from .ops import functional_sdp as FS
from .ops import kernels as K
from .ops import kernels_sdp as KS


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
        vl = Mo.crop_dev(x_lens)          # captured step: frames the reference's cropped batch does not have stay out of the taps
        for blk in self.conv:
            xs = Fn.conv1d(xs, blk[0].weight, blk[0].bias, act="relu", vlens=vl)
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
# VITS flow pieces: state_dict-compatible parameter holders; the arithmetic is csrc/sdp.hip + the shared GEMM /
# depthwise-conv kernels.  Everything is channel-last rows (B, T, C) in fp32 (the log-determinants of the flows
# are sums of many small terms; the reference computes them in fp32 as well).
# ---------------------------------------------------------------------------------------------------------
class _Transpose(nn.Module):
    """Placeholder keeping the reference's Sequential indices (flow.py:137-160); layout changes are free here."""

    def __init__(self, d1, d2):
        super().__init__()
        self.d1, self.d2 = d1, d2


class DilatedDepthSeparableConv(nn.Module):
    """flow.py:110-190: per layer  x = (x + dropout(gelu(LN(conv1x1(gelu(LN(dwconv_dilated(x*mask)))))))) ; final * mask."""

    def __init__(self, channels, kernel_size, layers, dropout_rate=0.0, eps=1e-5):
        super().__init__()
        self.dropout_rate = dropout_rate
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

    def forward(self, x, lens):
        """x (B, T, C), already zero past each utterance (so x == x*mask at every layer input) -> same, masked."""
        T = x.shape[1]
        p = self.dropout_rate if self.training else 0.0
        for blk in self.convs:
            dw, ln1, pw, ln2 = blk[0], blk[2], blk[5], blk[7]
            if FS._QUEUE_LN and FS.dw_ln_act_ok(x, dw.weight):                                  # one launch: conv -> LN -> GELU
                y, xr = FS.dw_ln_act(x, dw.weight, dw.bias, dw.dilation[0], ln1.weight, ln1.bias, ln1.eps, "gelu")
            else:
                if FS._QUEUE_LN:
                    y, xr = FA.dwconv1d_pass(x, dw.weight, dw.bias, dilation=dw.dilation[0])  # xr = x for the residual (one consumer of x)
                else:
                    y, xr = FA.dwconv1d(x, dw.weight, dw.bias, dilation=dw.dilation[0]), x
                y = FS.ln_act(y, ln1.weight, ln1.bias, ln1.eps, "gelu")
            y = Fn.linear(y, pw.weight, pw.bias)
            x = FS.ln_act(y, ln2.weight, ln2.bias, ln2.eps, "gelu", res=xr, lens=lens, T=T, p=p)
        return x


class ConvFlow(nn.Module):
    """flow.py:250-310: (xa, xb) -> (xa, RQ-spline(xb | DDS(input_conv(xa) + g)))."""

    def __init__(self, in_channels, hidden_channels, kernel_size, layers, bins=10, tail_bound=5.0):
        super().__init__()
        self.half_channels = in_channels // 2
        if self.half_channels != 1:
            raise NotImplementedError("ConvFlow is built for 2-channel flows (the duration predictor's)")
        self.hidden_channels, self.bins, self.tail_bound = hidden_channels, bins, tail_bound
        self.input_conv = nn.Conv1d(self.half_channels, hidden_channels, 1)
        self.dds_conv = DilatedDepthSeparableConv(hidden_channels, kernel_size, layers, dropout_rate=0.0)
        self.proj = nn.Conv1d(hidden_channels, self.half_channels * (bins * 3 - 1), 1)
        self.proj.weight.data.zero_()
        self.proj.bias.data.zero_()

    def params(self, xa, g, lens):
        h = FS.expand(xa, self.input_conv.weight, self.input_conv.bias, g, lens)
        h = self.dds_conv(h, lens)
        return Fn.linear(h, self.proj.weight, self.proj.bias)                 # (B, T, 3*bins-1); masked inside the spline

    def forward(self, xa, xb, g, lens, shared, which):
        h = self.params(xa, g, lens)
        return FS.spline(xb, h, 1.0 / math.sqrt(self.hidden_channels), self.tail_bound, lens, shared, which)

    def inverse(self, xa, xb, g, lens):
        h = self.params(xa, g, lens)
        out, _ = KS.rq_spline_fwd(xb.contiguous(), h.contiguous(), 1.0 / math.sqrt(self.hidden_channels), self.tail_bound, lens,
                                  inverse=True)
        return out


class ElementwiseAffineFlow(nn.Module):
    """flow.py:66-93 (parameter holder; applied inside the fused glue kernels)."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.register_parameter("m", nn.Parameter(torch.zeros(channels, 1)))
        self.register_parameter("logs", nn.Parameter(torch.zeros(channels, 1)))


class FlipFlow(nn.Module):
    """flow.py:18-34: channel flip == swapping the two (B, T) halves, free."""


class LogFlow(nn.Module):
    """flow.py:37-63 (inside the mid glue kernel)."""


class StochasticDurationPredictor(nn.Module):
    def __init__(self, channels=192, kernel_size=3, dropout_rate=0.5, flows=4, dds_conv_layers=3, global_channels=-1):
        super().__init__()
        if global_channels > 0:
            raise NotImplementedError("global conditioning is not used by AAS-VC (aas_vc.py:186: global_channels=-1)")
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
        self.noise = None   # set to a (B, 2, T) tensor to inject the draw (parity tests); consumed once

    def _randn(self, shape, device):
        if self.noise is not None:
            n, self.noise = self.noise.to(device=device, dtype=torch.float32).contiguous(), None
            return n
        return torch.randn(shape, device=device, dtype=torch.float32)

    def _condition(self, x, lens):
        """x (B, T, C) channel-last, detached (duration_predictor.py:230: no gradient to the encoder)."""
        x = K.cast(x.detach().contiguous(), torch.float32) if x.dtype != torch.float32 else x.detach()
        x = FS.mask_rows(Fn.linear(x, self.pre.weight, self.pre.bias), lens)
        x = self.dds(x, lens)
        return FS.mask_rows(Fn.linear(x, self.proj.weight, self.proj.bias), lens)

    def forward_cl(self, x, x_lens, w=None, inverse=False, noise_scale=1.0, normalize=False):
        """x (B, T, C) channel-last, x_lens: modules.Lens, w (B, T) durations -> NLL (B,) | durations (B, T).
        normalize: NLL / (number of non-padded text positions of the batch), the division of models/aas_vc.py:403 inside the kernel."""
        lens = x_lens.dev
        B, T, _ = x.shape
        x = self._condition(x, lens)
        if inverse:
            return self._inverse(x, lens, B, T, noise_scale)
        assert w is not None, "w must be provided."
        w = w.detach().float().contiguous()
        h_w = FS.expand(w, self.post_pre.weight, self.post_pre.bias, None, lens)
        h_w = FS.mask_rows(Fn.linear(self.post_dds(h_w, lens), self.post_proj.weight, self.post_proj.bias), lens)
        # gradient cut for the staged (data-parallel) backward pass: the two conditioning networks above -- input projection, DDS
        # stack, output projection, each -- are the last ~ 80 launches of this branch's backward pass and depend on nothing but the
        # gradients the flows accumulate in x and h_w: a stage plan may run them one stage later, beside the encoder's backward
        # pass (models/aas_vc.py: dp_plan).  Inactive (identity) outside distributed.OverlappedBackward.
        x, h_w = Fn.cut_point((x, h_w), "sdp_cond")              # both networks one stage later
        x = Fn.cut_point(x, "sdp_cond_x")                        # only the one behind x (the plan names ONE of the two cuts)
        # tensors with several consumers go through Fn.fan_out: their gradients are summed by one launch of this library at ONE autograd
        # node instead of by the engine's element-wise adds (16 per step here: x 5 x, g_q 4 x, every flow's pass-through half 2 x)
        n_q = sum(isinstance(f, ConvFlow) for f in self.post_flows[1:])
        n_p = sum(isinstance(f, ConvFlow) for f in self.flows[1:])
        x_add, *x_p = Fn.fan_out(x, 1 + n_p)
        g_q = Fn.fan_out(Fn.add_dropout(x_add, h_w, 0.0), n_q)
        noise = self._randn((B, 2, T), x.device)
        shared = FS.Shared()
        aff_q, aff_p = self.post_flows[0], self.flows[0]
        a, b = FS.head(noise, aff_q.m, aff_q.logs, lens)
        k = 0
        for flow in self.post_flows[1:]:
            if isinstance(flow, ConvFlow):            # ConvFlow then Flip: (a, b) -> (spline(b | a), a)
                a_in, a_next = Fn.fan_out(a, 2)
                a, b = flow(a_in, b, g_q[k], lens, shared, "q"), a_next
                k += 1
        zu, z1 = a, b
        zu, zu_tail = Fn.fan_out(zu, 2)
        a, b, lz = FS.mid(zu, z1, w, aff_p.m, aff_p.logs, lens)
        k = 0
        for flow in self.flows[1:]:
            if isinstance(flow, ConvFlow):
                a_in, a_next = Fn.fan_out(a, 2)
                a, b = flow(a_in, b, x_p[k], lens, shared, "p"), a_next
                k += 1
        return FS.tail(noise, zu_tail, lz, a, b, aff_q.logs, aff_p.logs, lens, shared, normalize)

    @torch.no_grad()
    def _inverse(self, x, lens, B, T, noise_scale):
        """duration_predictor.py:281-304: flows reversed, the first ConvFlow dropped, z = noise * noise_scale."""
        flows = list(reversed(self.flows))
        flows = flows[:-2] + [flows[-1]]
        z = self._randn((B, 2, T), x.device) * noise_scale
        a, b = z[:, 0].contiguous(), z[:, 1].contiguous()
        for flow in flows:
            if isinstance(flow, FlipFlow):
                a, b = b, a
            elif isinstance(flow, ConvFlow):
                b = flow.inverse(a, b, x, lens)
            else:                                         # ElementwiseAffine inverse + exp + ceil, channel 0 only
                return KS.sdp_inverse_out(a.contiguous(), lens, flow.m.detach().reshape(-1), flow.logs.detach().reshape(-1))
        raise RuntimeError("flow list must end with the affine flow")

    def forward(self, x, x_mask, w=None, g=None, inverse=False, noise_scale=1.0):
        """Reference signature (duration_predictor.py:211-229): x (B, C, T), x_mask (B, 1, T), w (B, 1, T)."""
        if g is not None:
            raise NotImplementedError("global conditioning is not used by AAS-VC")
        lens = Mo.Lens.of(x_mask.reshape(x_mask.shape[0], -1).sum(dim=1).long(), x.device)
        out = self.forward_cl(x.transpose(1, 2), lens, None if w is None else w.reshape(w.shape[0], -1), inverse, noise_scale)
        return out.unsqueeze(1) if inverse else out
