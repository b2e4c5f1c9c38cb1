"""Conformer encoder (also used as the AAS-VC "decoder") on the HIP kernels.

Reference: seq2seq_vc/modules/conformer/encoder.py:72-293, encoder_layer.py:44-179, convolution.py:13-79.
Same constructor arguments and state_dict keys; padded frames are NOT masked inside the convolution
module or its BatchNorm, exactly like the reference.
"""
import torch
from torch import nn

from . import modules as Mo
from .ops import functional as Fn
from .ops import functional_aas as FA


class ConvolutionModule(nn.Module):
    """PW-conv (d->2d) -> GLU -> depthwise conv k -> BatchNorm1d -> Swish -> PW-conv (convolution.py:56-79)."""

    def __init__(self, channels, kernel_size, activation="swish", bias=True):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, kernel_size=1, stride=1, padding=0, bias=bias)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, stride=1, padding=(kernel_size - 1) // 2,
                                        groups=channels, bias=bias)
        self.norm = nn.BatchNorm1d(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, kernel_size=1, stride=1, padding=0, bias=bias)
        self.activation = activation

    def forward(self, x, lens=None):
        """lens: the Lens of x.  In a captured step on a batch shorter than its padded shape (modules.LensBank) its crop() marks the
        frames the reference's tensor does not have: zero padding for the depthwise convolution, outside the BatchNorm statistics."""
        c1, c2, bn = self.pointwise_conv1, self.pointwise_conv2, self.norm
        vl = Mo.crop_dev(lens) if self.training else None
        y = Fn.linear(x, c1.weight, c1.bias)           # 1x1 conv == Linear on channel-last ((N,K,1) weight accepted)
        if FA.convmod_core_ok(y, self.depthwise_conv.weight, self.training, self.activation):
            # bf16 training: GLU -> depthwise conv -> batch statistics -> BatchNorm + Swish on the fused kernels (csrc/convmod.hip)
            y = FA.convmod_core(y, self.depthwise_conv.weight, self.depthwise_conv.bias, bn.weight, bn.bias, bn.running_mean,
                                bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum, vlens=vl)
            return Fn.linear(y, c2.weight, c2.bias)
        y = Fn.crop_rows(Fn.glu(y), vl)
        y = FA.dwconv1d(y, self.depthwise_conv.weight, self.depthwise_conv.bias)
        y = Fn.batch_norm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, self.training,
                              self.activation, 0.0, bn.eps, bn.momentum, vlens=vl)
        return Fn.linear(y, c2.weight, c2.bias)


class EncoderLayer(nn.Module):
    """Macaron FFN -> (rel-pos) MHA -> conv module -> FFN -> norm_final (conformer/encoder_layer.py:79-179)."""

    def __init__(self, size, self_attn, feed_forward, feed_forward_macaron, conv_module, dropout_rate, normalize_before=True,
                 concat_after=False, stochastic_depth_rate=0.0):
        super().__init__()
        if concat_after or stochastic_depth_rate > 0:
            raise NotImplementedError("concat_after / stochastic depth are not used by any recipe")
        self.self_attn, self.feed_forward = self_attn, feed_forward
        self.feed_forward_macaron, self.conv_module = feed_forward_macaron, conv_module
        self.norm_ff = Mo.LayerNorm(size)
        self.norm_mha = Mo.LayerNorm(size)
        if feed_forward_macaron is not None:
            self.norm_ff_macaron = Mo.LayerNorm(size)
            self.ff_scale = 0.5
        else:
            self.ff_scale = 1.0
        if conv_module is not None:
            self.norm_conv = Mo.LayerNorm(size)
            self.norm_final = Mo.LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.size, self.normalize_before = size, normalize_before

    def _attn(self, y, pos_emb, klens):
        if pos_emb is not None:
            return self.self_attn(y, y, y, pos_emb, klens)
        return self.self_attn(y, y, y, klens)

    def forward(self, x, pos_emb, klens):
        p = self.dropout_rate if self.training else 0.0
        pre = self.normalize_before
        # every "x = res + scale*dropout(h)" is fused with the LayerNorm that follows it
        steps = []
        def ff(mod):       # the Conv1d feed-forward mixes along time: it needs the lengths (modules.MultiLayeredConv1d)
            return (lambda y: mod(y, klens)) if isinstance(mod, Mo.MultiLayeredConv1d) else (lambda y: mod(y))

        if self.feed_forward_macaron is not None:
            steps.append((self.norm_ff_macaron, ff(self.feed_forward_macaron), self.ff_scale))
        steps.append((self.norm_mha, lambda y: self._attn(y, pos_emb, klens), 1.0))
        if self.conv_module is not None:
            steps.append((self.norm_conv, lambda y: self.conv_module(y, klens), 1.0))
        steps.append((self.norm_ff, ff(self.feed_forward), self.ff_scale))
        if pre:
            n0 = steps[0][0]          # x feeds this norm AND the first residual: the residual takes the norm's pass-through alias
            y, x = Fn.layer_norm(x, n0.weight, n0.bias, n0.eps, passthrough=True)
            for i, (_, fn, scale) in enumerate(steps):
                h = fn(y)
                nxt = steps[i + 1][0] if i + 1 < len(steps) else (self.norm_final if self.conv_module is not None else None)
                if nxt is not None:
                    y, x = Fn.add_dropout_layer_norm(x, h, nxt.weight, nxt.bias, nxt.eps, p, scale)
                else:
                    x = Fn.add_dropout(x, h, p, scale)
                    y = x
            return y if self.conv_module is not None else x
        for norm, fn, scale in steps:
            x, _ = Fn.add_dropout_layer_norm(x, fn(x), norm.weight, norm.bias, norm.eps, p, scale)
        if self.conv_module is not None:
            x = self.norm_final(x)
        return x


class ConformerEncoder(nn.Module):
    def __init__(self, idim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                 positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d", normalize_before=True,
                 concat_after=False, positionwise_layer_type="linear", positionwise_conv_kernel_size=3, macaron_style=False,
                 pos_enc_layer_type="abs_pos", selfattention_layer_type="selfattn", use_cnn_module=False, zero_triu=False,
                 cnn_module_kernel=31, padding_idx=-1, stochastic_depth_rate=0.0, intermediate_layers=None, ctc_softmax=None,
                 conditioning_layer_dim=None):
        super().__init__()
        if intermediate_layers is not None or ctc_softmax is not None:
            raise NotImplementedError("intermediate CTC branches are not on the VC hot path")
        if pos_enc_layer_type == "abs_pos":
            pos_enc_class = Mo.PositionalEncoding
        elif pos_enc_layer_type == "scaled_abs_pos":
            pos_enc_class = Mo.ScaledPositionalEncoding
        elif pos_enc_layer_type == "rel_pos":
            assert selfattention_layer_type == "rel_selfattn"
            pos_enc_class = Mo.RelPositionalEncoding
        elif pos_enc_layer_type == "legacy_rel_pos":
            assert selfattention_layer_type == "legacy_rel_selfattn"
            pos_enc_class = Mo.LegacyRelPositionalEncoding
        else:
            raise ValueError("unknown pos_enc_layer: " + pos_enc_layer_type)
        self.input_layer = input_layer
        if input_layer == "linear":
            self.embed = nn.Sequential(nn.Linear(idim, attention_dim), nn.LayerNorm(attention_dim), nn.Dropout(dropout_rate),
                                       pos_enc_class(attention_dim, positional_dropout_rate))
        elif input_layer == "conv2d":
            self.embed = Mo.Conv2dSubsampling(idim, attention_dim, dropout_rate, pos_enc_class(attention_dim, positional_dropout_rate))
        elif input_layer is None:
            self.embed = nn.Sequential(pos_enc_class(attention_dim, positional_dropout_rate))
        else:
            raise ValueError("unsupported input_layer: " + str(input_layer))
        self.normalize_before = normalize_before
        if selfattention_layer_type == "selfattn":
            att = lambda: Mo.MultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate)
        elif selfattention_layer_type == "legacy_rel_selfattn":
            att = lambda: Mo.LegacyRelPositionMultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate)
        elif selfattention_layer_type == "rel_selfattn":
            att = lambda: Mo.RelPositionMultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate, zero_triu)
        else:
            raise ValueError("unknown encoder_attn_layer: " + selfattention_layer_type)
        if positionwise_layer_type == "linear":
            ff = lambda: Mo.PositionwiseFeedForward(attention_dim, linear_units, dropout_rate, "swish")
        elif positionwise_layer_type == "conv1d":
            ff = lambda: Mo.MultiLayeredConv1d(attention_dim, linear_units, positionwise_conv_kernel_size, dropout_rate)
        else:
            raise NotImplementedError("Support only linear or conv1d.")
        self.encoders = Mo.MultiSequential(*[
            EncoderLayer(attention_dim, att(), ff(), ff() if macaron_style else None,
                         ConvolutionModule(attention_dim, cnn_module_kernel, "swish") if use_cnn_module else None,
                         dropout_rate, normalize_before, concat_after) for _ in range(num_blocks)])
        if normalize_before:
            self.after_norm = Mo.LayerNorm(attention_dim)
        self.dropout_rate = dropout_rate

    def forward(self, xs, lens, exact_lens=False):
        """xs (B,T,idim) compute dtype; lens: Lens or None -> (ys (B,T',adim), lens')."""
        if self.input_layer == "conv2d":
            xs, lens = self.embed(xs, lens, exact_lens)   # Conv2dSubsampling applies its positional encoding itself
        elif self.input_layer == "linear":
            lin, ln = self.embed[0], self.embed[1]
            xs = Fn.linear(xs, lin.weight, lin.bias)
            xs = Fn.layer_norm(xs, ln.weight, ln.bias, ln.eps)
            xs = Fn.dropout(xs, self.embed[2].p, self.training)
            xs = self.embed[3](xs)
        else:
            xs = self.embed[0](xs)
        pos_emb = None
        if isinstance(xs, tuple):
            xs, pos_emb = xs
        cut_name = getattr(self, "cut_name", None)      # data-parallel overlap: "<name>.<i>" cuts the graph at the input of layer i
        for li, layer in enumerate(self.encoders):
            if cut_name is not None and li > 0:
                xs = Fn.cut_point(xs, f"{cut_name}.{li}")
            xs = layer(xs, pos_emb, lens)
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, lens
