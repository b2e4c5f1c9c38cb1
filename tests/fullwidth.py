"""Full-width single-layer parity cases (VERDICT r1 "weak #3"): the layer shapes the benchmark configurations run
-- d=384 / 4 heads (d_k 96) Transformer encoder + decoder layers of VTN vc1, d=384 / 2 heads (d_k 192) and
d=1536 / 2 heads (d_k 768) Conformer layers with kernel 15 of AAS-VC vc2 -- pinned to the REFERENCE's own layer classes.

The weights are too large to commit (28 M parameters for the d=1536 layer), so both sides regenerate them from a seed
with `seeded_state` (torch CPU generator: bit-identical wherever this torch build runs); the fixture
(tests/golden/fw_*.npz, written by tools/gen_golden.py from the imported reference) holds a float64 checksum of every
generated tensor, the inputs, the reference's output, its input gradient and strided samples of its parameter gradients.
"""
import math

import torch

CASES = {
    # name: kind, d, heads, ffn units, conv kernel, B, T (T_mem for the decoder layer), valid lengths
    "fw_enc384": dict(kind="encoder", d=384, h=4, units=1536, B=3, T=63, lens=[63, 50, 33], pre_ln=True, seed=401),
    "fw_dec384": dict(kind="decoder", d=384, h=4, units=1536, B=3, T=64, lens=[64, 41, 27], Tm=63, mlens=[63, 50, 33],
                      pre_ln=False, seed=402),
    "fw_conf384": dict(kind="conformer", d=384, h=2, units=1536, k=15, B=2, T=96, lens=[96, 71], pre_ln=True, seed=403),
    "fw_conf1536": dict(kind="conformer", d=1536, h=2, units=1536, k=15, B=2, T=64, lens=[64, 45], pre_ln=True, seed=404),
}


def grad_stride(numel):
    """Parameter gradients are stored as flat[::stride]: every 7th element of small tensors, ~16 K samples of big ones
    (odd stride, so that the samples walk through all columns of a power-of-two-wide matrix)."""
    return max(7, numel // 16384) | 1


def seeded_state(named_shapes, seed):
    """name -> tensor for every (name, shape) in order: matrices / conv kernels ~ N(0, 1/fan_in), LayerNorm / BatchNorm
    scales 1 + 0.1 N, biases and position biases 0.1 N.  BatchNorm buffers keep their defaults (not listed)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        n = torch.randn(shape, generator=g)
        leaf = name.rsplit(".", 1)[-1]
        if len(shape) >= 2 and leaf == "weight":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = n / math.sqrt(fan_in)
        elif leaf == "weight":                     # LayerNorm / BatchNorm scale
            t = 1.0 + 0.1 * n
        else:                                      # biases, pos_bias_u / pos_bias_v
            t = 0.1 * n
        out[name] = t
    return out


def checksum(t):
    """Exact, order-independent checksum of an fp32 tensor: weighted int64 sum of the bit patterns."""
    b = t.detach().float().contiguous().reshape(-1).view(torch.int32).long()
    w = torch.arange(1, b.numel() + 1, dtype=torch.int64) % 97 + 1
    return int((b * w).sum())


def inputs(c):
    """x (B,T,d), memory (decoder only), upstream gradient dy (B,T,d) -- unit-variance activations."""
    g = torch.Generator().manual_seed(c["seed"] + 1000)
    x = torch.randn(c["B"], c["T"], c["d"], generator=g)
    dy = torch.randn(c["B"], c["T"], c["d"], generator=g) / math.sqrt(c["d"])
    mem = torch.randn(c["B"], c["Tm"], c["d"], generator=g) if c["kind"] == "decoder" else None
    return x, mem, dy


def layer_param_shapes(c):
    """(name, shape) of every parameter of the layer, in the reference's named_parameters() order."""
    d, u = c["d"], c["units"]

    def mha(p, rel=False):
        out = []
        if rel:
            pass
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            out += [(f"{p}.{n}.weight", (d, d)), (f"{p}.{n}.bias", (d,))]
        return out

    def ffn(p):
        return [(f"{p}.w_1.weight", (u, d)), (f"{p}.w_1.bias", (u,)), (f"{p}.w_2.weight", (d, u)), (f"{p}.w_2.bias", (d,))]

    def ln(p):
        return [(f"{p}.weight", (d,)), (f"{p}.bias", (d,))]

    if c["kind"] == "encoder":
        return mha("self_attn") + ffn("feed_forward") + ln("norm1") + ln("norm2")
    if c["kind"] == "decoder":
        return mha("self_attn") + mha("src_attn") + ffn("feed_forward") + ln("norm1") + ln("norm2") + ln("norm3")
    h, k = c["h"], c["k"]
    att = [("self_attn.pos_bias_u", (h, d // h)), ("self_attn.pos_bias_v", (h, d // h))] + mha("self_attn") + \
          [("self_attn.linear_pos.weight", (d, d))]
    conv = [("conv_module.pointwise_conv1.weight", (2 * d, d, 1)), ("conv_module.pointwise_conv1.bias", (2 * d,)),
            ("conv_module.depthwise_conv.weight", (d, 1, k)), ("conv_module.depthwise_conv.bias", (d,)),
            ("conv_module.norm.weight", (d,)), ("conv_module.norm.bias", (d,)),
            ("conv_module.pointwise_conv2.weight", (d, d, 1)), ("conv_module.pointwise_conv2.bias", (d,))]
    return att + ffn("feed_forward") + ffn("feed_forward_macaron") + conv + ln("norm_ff") + ln("norm_mha") + \
        ln("norm_ff_macaron") + ln("norm_conv") + ln("norm_final")


def masks(c):
    T = c["T"]
    key_mask = (torch.arange(T)[None, :] < torch.tensor(c["lens"])[:, None]).unsqueeze(1)          # (B,1,T) True = valid
    if c["kind"] != "decoder":
        return key_mask, None, None
    mem_mask = (torch.arange(c["Tm"])[None, :] < torch.tensor(c["mlens"])[:, None]).unsqueeze(1)
    tgt_mask = key_mask & torch.tril(torch.ones(T, T, dtype=torch.bool))[None]
    return key_mask, tgt_mask, mem_mask


def oracle_layer(c, sd, x, mem):
    """The CPU oracle's restatement of the layer (oracle/nets.py) on a flat state dict."""
    from oracle import nets as N
    rt = N.Runtime(True, False)
    key_mask, tgt_mask, mem_mask = masks(c)
    p = N.P(sd, "")
    if c["kind"] == "encoder":
        return N.encoder_layer(p, x, key_mask, c["h"], rt, 0.0, 0.0, c["pre_ln"], "l")
    if c["kind"] == "decoder":
        return N.decoder_layer(p, x, tgt_mask, mem, mem_mask, c["h"], rt, 0.0, c["pre_ln"], "l")
    xs, pe = N.rel_posenc(x, rt, 0.0)
    return N.conformer_layer(p, xs, pe, key_mask, c["h"], rt, 0.0, 0.0, c["pre_ln"], False, "l")
