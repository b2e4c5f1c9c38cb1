"""Functional fp32 CPU restatement of the seq2seq-vc building blocks -- TEST INFRASTRUCTURE ONLY.

Everything is a pure function over a flat `state_dict` (reference key names) addressed through the
`P` prefix view, written with torch.nn.functional on CPU tensors.  Each function cites the reference
lines it restates (paths relative to the reference root).  Dropout is applied with torch's CPU RNG
when `rt.drop` is true (CPU-baseline timing) and skipped otherwise (parity runs).
"""
import math

import torch
import torch.nn.functional as F


class P:
    """Prefix view over a flat state_dict: P(sd, 'encoder.')['after_norm.weight']."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def get(self, k, default=None):
        return self.sd.get(self.prefix + k, default)

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, name):
        return P(self.sd, self.prefix + name + ".")


class Runtime:
    """Per-call switches: training (BatchNorm batch stats), drop (apply dropout), captured attention maps."""

    def __init__(self, training=True, drop=False):
        self.training, self.drop = training, drop
        self.attn = {}

    def dropout(self, x, p, always=False):
        if p <= 0.0 or not self.drop:
            return x
        if always or self.training:
            return F.dropout(x, p, training=True)
        return x


# ------------------------------------------------------------------ masks (layers/utils.py:93-121, mask.py:9-22)
def non_pad_mask(lens, maxlen=None):
    lens = torch.as_tensor(lens, dtype=torch.long)
    maxlen = int(lens.max()) if maxlen is None else maxlen
    return torch.arange(maxlen)[None, :] < lens[:, None]


def pad_mask(lens, maxlen=None):
    return ~non_pad_mask(lens, maxlen)


def causal_mask(n):
    return torch.tril(torch.ones(n, n, dtype=torch.bool))


# ------------------------------------------------------------------ small layers
def linear(p, x):
    return F.linear(x, p["weight"], p.get("bias"))


def layer_norm(p, x, eps=1e-12):  # modules/transformer/layer_norm.py:12-42 (eps 1e-12)
    return F.layer_norm(x, (x.shape[-1],), p["weight"], p["bias"], eps)


def sin_table(n, d, reverse=False):  # layers/positional_encoding.py:35-55
    pos = torch.arange(n - 1, -1, -1.0) if reverse else torch.arange(0, n, dtype=torch.float32)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos[:, None] * div)
    pe[:, 1::2] = torch.cos(pos[:, None] * div)
    return pe


def rel_table(n, d):  # layers/positional_encoding.py:261-291 -> rows for relative offsets n-1 .. -(n-1)
    pos = torch.arange(0, n, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    plus, minus = torch.zeros(n, d), torch.zeros(n, d)
    plus[:, 0::2], plus[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    minus[:, 0::2], minus[:, 1::2] = torch.sin(-pos * div), torch.cos(-pos * div)
    return torch.cat([torch.flip(plus, [0]), minus[1:]], dim=0)  # (2n-1, d)


def scaled_posenc(p, x, rt, drop_p):  # positional_encoding.py:94-106
    pe = sin_table(x.shape[1], x.shape[2])
    return rt.dropout(x + p["alpha"] * pe[None], drop_p)


def abs_posenc(x, rt, drop_p):  # positional_encoding.py:57-70
    pe = sin_table(x.shape[1], x.shape[2])
    return rt.dropout(x * math.sqrt(x.shape[2]) + pe[None], drop_p)


def rel_posenc(x, rt, drop_p):  # positional_encoding.py:293-309 -> (x*sqrt(d), pos_emb (1, 2T-1, d))
    pe = rel_table(x.shape[1], x.shape[2])[None].to(x.dtype)      # the reference keeps its table in x's dtype (positional_encoding.py:281-291)
    return rt.dropout(x * math.sqrt(x.shape[2]), drop_p), rt.dropout(pe, drop_p)


def legacy_rel_posenc(x, rt, drop_p, max_len=5000):
    """positional_encoding.py:198-235.  The reference builds the REVERSED table once for max_len=5000
    (positions 4999..0) and slices its first T rows, so the embedding handed to the attention is for
    positions 4999, 4998, ... -- restated as is."""
    n = max(max_len, x.shape[1])
    pe = sin_table(n, x.shape[2], reverse=True)[: x.shape[1]][None].to(x.dtype)
    return rt.dropout(x * math.sqrt(x.shape[2]), drop_p), rt.dropout(pe, drop_p)


# ------------------------------------------------------------------ attention (modules/transformer/attention.py)
def _heads(x, h):
    b, t, d = x.shape
    return x.view(b, t, h, d // h).transpose(1, 2)


def _attend(p, v, scores, mask, rt, drop_p, name):  # attention.py:63-93
    if mask is not None:
        m = ~mask.unsqueeze(1)
        scores = scores.masked_fill(m, torch.finfo(scores.dtype).min)
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    if name is not None:
        rt.attn[name] = attn
    ctx = torch.matmul(rt.dropout(attn, drop_p), v)
    b, h, t, dk = ctx.shape
    return linear(p.sub("linear_out"), ctx.transpose(1, 2).reshape(b, t, h * dk))


def mha(p, q, k, v, mask, h, rt, drop_p=0.0, name=None):  # attention.py:39-111
    qh, kh, vh = _heads(linear(p.sub("linear_q"), q), h), _heads(linear(p.sub("linear_k"), k), h), _heads(linear(p.sub("linear_v"), v), h)
    scores = torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(qh.shape[-1])
    return _attend(p, vh, scores, mask, rt, drop_p, name)


def rel_shift(x, legacy):  # attention.py:237-260 (new, cropped) / :142-160 (legacy)
    b, h, t, l = x.shape
    xp = torch.cat([x.new_zeros(b, h, t, 1), x], dim=-1).view(b, h, l + 1, t)[:, :, 1:].reshape(b, h, t, l)
    return xp if legacy else xp[..., : l // 2 + 1]


def rel_mha(p, x_q, x_kv, pos_emb, mask, h, rt, drop_p=0.0, legacy=False, name=None):  # attention.py:262-305 / :162-206
    q, kh, vh = linear(p.sub("linear_q"), x_q), _heads(linear(p.sub("linear_k"), x_kv), h), _heads(linear(p.sub("linear_v"), x_kv), h)
    b, t, d = q.shape
    dk = d // h
    q4 = q.view(b, t, h, dk)
    ph = F.linear(pos_emb, p["linear_pos.weight"]).view(pos_emb.shape[0], -1, h, dk).transpose(1, 2)
    qu = (q4 + p["pos_bias_u"]).transpose(1, 2)
    qv = (q4 + p["pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(qu, kh.transpose(-2, -1))
    bd = rel_shift(torch.matmul(qv, ph.transpose(-2, -1)), legacy)
    return _attend(p, vh, (ac + bd) / math.sqrt(dk), mask, rt, drop_p, name)


# ------------------------------------------------------------------ feed-forward (positionwise_feed_forward.py:30-32)
def swish(x):
    return x * torch.sigmoid(x)


def ffn(p, x, rt, drop_p, act=torch.relu):
    return linear(p.sub("w_2"), rt.dropout(act(linear(p.sub("w_1"), x)), drop_p))


def ffn_conv1d(p, x, rt, drop_p):  # multi_layer_conv.py:52-63
    k = p["w_1.weight"].shape[-1]
    hdn = torch.relu(F.conv1d(x.transpose(1, 2), p["w_1.weight"], p["w_1.bias"], padding=(k - 1) // 2)).transpose(1, 2)
    return F.conv1d(rt.dropout(hdn, drop_p).transpose(1, 2), p["w_2.weight"], p["w_2.bias"], padding=(k - 1) // 2).transpose(1, 2)


# ------------------------------------------------------------------ Transformer layers
def encoder_layer(p, x, mask, h, rt, drop_p, attn_p, pre_ln, name):  # encoder_layer.py:61-119 (no cache, no concat)
    res = x
    y = layer_norm(p.sub("norm1"), x) if pre_ln else x
    x = res + rt.dropout(mha(p.sub("self_attn"), y, y, y, mask, h, rt, attn_p, name + ".self_attn"), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm1"), x)
    res = x
    y = layer_norm(p.sub("norm2"), x) if pre_ln else x
    x = res + rt.dropout(ffn(p.sub("feed_forward"), y, rt, drop_p), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm2"), x)
    return x


def decoder_layer(p, x, tgt_mask, mem, mem_mask, h, rt, drop_p, pre_ln, name):  # decoder_layer.py:63-134
    res = x
    y = layer_norm(p.sub("norm1"), x) if pre_ln else x
    x = res + rt.dropout(mha(p.sub("self_attn"), y, y, y, tgt_mask, h, rt, 0.0, name + ".self_attn"), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm1"), x)
    res = x
    y = layer_norm(p.sub("norm2"), x) if pre_ln else x
    x = res + rt.dropout(mha(p.sub("src_attn"), y, mem, mem, mem_mask, h, rt, 0.0, name + ".src_attn"), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm2"), x)
    res = x
    y = layer_norm(p.sub("norm3"), x) if pre_ln else x
    x = res + rt.dropout(ffn(p.sub("feed_forward"), y, rt, drop_p), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm3"), x)
    return x


def batch_norm(p, x, rt, eps=1e-5, momentum=0.1):
    """x: (B, C, T); train mode uses batch statistics over (B, T) incl. padded frames
    and updates the running buffers in place, eval mode uses the running buffers."""
    if rt.training and p.has("num_batches_tracked"):
        p["num_batches_tracked"].add_(1)
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], rt.training, momentum, eps)


def conv_module(p, x, rt):  # modules/conformer/convolution.py:56-79
    k = p["depthwise_conv.weight"].shape[-1]
    y = F.conv1d(x.transpose(1, 2), p["pointwise_conv1.weight"], p["pointwise_conv1.bias"])
    y = F.glu(y, dim=1)
    y = F.conv1d(y, p["depthwise_conv.weight"], p["depthwise_conv.bias"], padding=(k - 1) // 2, groups=y.shape[1])
    y = swish(batch_norm(p.sub("norm"), y, rt))
    return F.conv1d(y, p["pointwise_conv2.weight"], p["pointwise_conv2.bias"]).transpose(1, 2)


def conformer_layer(p, x, pos_emb, mask, h, rt, drop_p, attn_p, pre_ln, legacy, name, ff="linear"):
    """modules/conformer/encoder_layer.py:79-179 (macaron + rel-pos MHA + conv module + FFN + norm_final)."""
    feed = (lambda q, y: ffn(q, y, rt, drop_p, act=swish)) if ff == "linear" else (lambda q, y: ffn_conv1d(q, y, rt, drop_p))
    macaron = p.has("feed_forward_macaron.w_1.weight")
    scale = 0.5 if macaron else 1.0
    if macaron:
        res = x
        y = layer_norm(p.sub("norm_ff_macaron"), x) if pre_ln else x
        x = res + scale * rt.dropout(feed(p.sub("feed_forward_macaron"), y), drop_p)
        if not pre_ln:
            x = layer_norm(p.sub("norm_ff_macaron"), x)
    res = x
    y = layer_norm(p.sub("norm_mha"), x) if pre_ln else x
    if pos_emb is not None:
        a = rel_mha(p.sub("self_attn"), y, y, pos_emb, mask, h, rt, attn_p, legacy, name + ".self_attn")
    else:
        a = mha(p.sub("self_attn"), y, y, y, mask, h, rt, attn_p, name + ".self_attn")
    x = res + rt.dropout(a, drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm_mha"), x)
    has_conv = p.has("conv_module.pointwise_conv1.weight")
    if has_conv:
        res = x
        y = layer_norm(p.sub("norm_conv"), x) if pre_ln else x
        x = res + rt.dropout(conv_module(p.sub("conv_module"), y, rt), drop_p)
        if not pre_ln:
            x = layer_norm(p.sub("norm_conv"), x)
    res = x
    y = layer_norm(p.sub("norm_ff"), x) if pre_ln else x
    x = res + scale * rt.dropout(feed(p.sub("feed_forward"), y), drop_p)
    if not pre_ln:
        x = layer_norm(p.sub("norm_ff"), x)
    if has_conv:
        x = layer_norm(p.sub("norm_final"), x)
    return x


# ------------------------------------------------------------------ input layers
def conv2d_subsample(p, x, mask):  # modules/transformer/subsampling.py:74-94 (without the trailing pos-enc)
    y = torch.relu(F.conv2d(x.unsqueeze(1), p["conv.0.weight"], p["conv.0.bias"], stride=2))
    y = torch.relu(F.conv2d(y, p["conv.2.weight"], p["conv.2.bias"], stride=2))
    b, c, t, f = y.shape
    y = y.transpose(1, 2).reshape(b, t, c * f)
    w = p["out.0.weight"] if p.has("out.0.weight") else p["out.weight"]
    bb = p["out.0.bias"] if p.has("out.0.bias") else p["out.bias"]
    y = F.linear(y, w, bb)
    return y, (None if mask is None else mask[:, :, :-2:2][:, :, :-2:2])


def prenet(p, x, rt, drop_p):  # modules/pre_postnets.py:53-66 (dropout always on)
    i = 0
    while p.has(f"prenet.{i}.0.weight"):
        x = rt.dropout(torch.relu(linear(p.sub(f"prenet.{i}.0"), x)), drop_p, always=True)
        i += 1
    return x


def postnet(p, x, rt, drop_p=0.5):  # modules/pre_postnets.py:108-185 ; x: (B, odim, T)
    n = 0
    while p.has(f"postnet.{n}.0.weight"):
        n += 1
    for i in range(n):
        w = p[f"postnet.{i}.0.weight"]
        x = F.conv1d(x, w, None, padding=(w.shape[-1] - 1) // 2)
        if p.has(f"postnet.{i}.1.weight"):
            x = batch_norm(p.sub(f"postnet.{i}.1"), x, rt)
        if i < n - 1:
            x = torch.tanh(x)
        x = rt.dropout(x, drop_p)
    return x


# ------------------------------------------------------------------ encoder / decoder stacks
def n_layers(p, stem):
    n = 0
    while any(k.startswith(p.prefix + f"{stem}.{n}.") for k in p.sd):
        n += 1
    return n


def transformer_encoder(p, xs, mask, c, rt, embed):
    """modules/transformer/encoder.py:283-329.  embed: 'conv2d-scaled' (VTN) or 'embed-scaled' (TTS)."""
    dp = c.get("enc_dropout", 0.1)
    pdp = c.get("enc_pos_dropout", 0.1)
    if embed == "conv2d-scaled":
        x, mask = conv2d_subsample(p.sub("embed"), xs, mask)
        x = scaled_posenc(p.sub("embed.out.1"), x, rt, pdp)
    else:  # Embedding (padding_idx 0) + scaled pos-enc: models/transformer_tts.py:63-77
        x = F.embedding(xs, p["embed.0.weight"], padding_idx=0)
        x = scaled_posenc(p.sub("embed.1"), x, rt, pdp)
    pre = c.get("encoder_normalize_before", True)
    for i in range(n_layers(p, "encoders")):
        x = encoder_layer(p.sub(f"encoders.{i}"), x, mask, c["aheads"], rt, dp, 0.0, pre, f"encoder.encoders.{i}")
    if pre:
        x = layer_norm(p.sub("after_norm"), x)
    return x, mask


def conformer_encoder(p, xs, mask, c, rt, heads, input_layer, pos_type, dp, pdp, adp, pre, name, ff="linear"):
    """modules/conformer/encoder.py:237-293."""
    legacy = pos_type == "legacy_rel_pos"

    def pos(x):
        if pos_type == "rel_pos":
            return rel_posenc(x, rt, pdp)
        if pos_type == "legacy_rel_pos":
            return legacy_rel_posenc(x, rt, pdp)
        if pos_type == "scaled_abs_pos":
            return scaled_posenc(p.sub("embed.%d" % (3 if input_layer == "linear" else 0)), x, rt, pdp), None
        return abs_posenc(x, rt, pdp), None

    if input_layer == "linear":  # conformer/encoder.py:117-123: Linear -> LayerNorm(1e-5) -> Dropout -> pos-enc
        x = linear(p.sub("embed.0"), xs)
        x = F.layer_norm(x, (x.shape[-1],), p["embed.1.weight"], p["embed.1.bias"], 1e-5)
        x, pe = pos(rt.dropout(x, dp))
    elif input_layer == "conv2d":
        x, mask = conv2d_subsample(p.sub("embed"), xs, mask)
        x, pe = pos(x)
    else:  # None: pos-enc only
        x, pe = pos(xs)
    for i in range(n_layers(p, "encoders")):
        x = conformer_layer(p.sub(f"encoders.{i}"), x, pe, mask, heads, rt, dp, adp, pre, legacy, f"{name}.encoders.{i}", ff)
    if pre:
        x = layer_norm(p.sub("after_norm"), x)
    return x, mask


def transformer_decoder(p, ys_in, tgt_mask, mem, mem_mask, c, rt):
    """modules/transformer/decoder.py:207-237 with input layer Sequential(Prenet, Linear) + scaled pos-enc."""
    x = prenet(p.sub("embed.0.0"), ys_in, rt, c.get("dprenet_dropout_rate", 0.5))
    x = linear(p.sub("embed.0.1"), x)
    x = scaled_posenc(p.sub("embed.1"), x, rt, 0.1)
    pre = c.get("decoder_normalize_before", False)
    for i in range(n_layers(p, "decoders")):
        x = decoder_layer(p.sub(f"decoders.{i}"), x, tgt_mask, mem, mem_mask, c["aheads"], rt, 0.1, pre, f"decoder.decoders.{i}")
    if pre:
        x = layer_norm(p.sub("after_norm"), x)
    return x
