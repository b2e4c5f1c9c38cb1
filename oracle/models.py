"""Functional fp32 CPU restatement of the seq2seq-vc models and losses -- TEST INFRASTRUCTURE ONLY.

`cfg` dicts use the reference constructor keyword names (models/vtn.py:15-62, models/aas_vc.py:39-111,
models/transformer_tts.py:14-43).  State lives in a plain state_dict with the reference key names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nets as N
from .mas import viterbi_decode as _viterbi_np
from .nets import P, Runtime


def _cfg(c, k, d):
    v = c.get(k, d)
    return d if v is None else v


# =============================================================================================
# VTN  (models/vtn.py:207-300 forward, :302-394 inference)
# =============================================================================================
def _vtn_encoder(sd, c, xs, x_masks, rt):
    if c.get("encoder_type", "transformer") == "transformer":
        cc = dict(c, enc_dropout=_cfg(c, "transformer_enc_dropout_rate", 0.1), enc_pos_dropout=0.1)
        return N.transformer_encoder(P(sd, "encoder."), xs, x_masks, cc, rt, "conv2d-scaled")
    pos_type = c.get("conformer_pos_enc_layer_type", "rel_pos")
    if c.get("conformer_rel_pos_type", "legacy") == "legacy" and pos_type == "rel_pos":
        pos_type = "legacy_rel_pos"  # vtn.py:83-99 fallback
    return N.conformer_encoder(P(sd, "encoder."), xs, x_masks, c, rt, c["aheads"], "conv2d", pos_type,
                               _cfg(c, "transformer_enc_dropout_rate", 0.1),
                               _cfg(c, "transformer_enc_positional_dropout_rate", 0.1),
                               _cfg(c, "transformer_enc_attn_dropout_rate", 0.1),
                               c.get("encoder_normalize_before", True), "encoder",
                               c.get("positionwise_layer_type", "linear"))


def _ar_teacher_forced(sd, c, hs, hs_masks, ys, labels, olens, rt):
    """Shared tail of VTN.forward / TransformerTTS.forward (vtn.py:227-274 == transformer_tts.py:160-203)."""
    r = _cfg(c, "decoder_reduction_factor", 2)
    odim = c["odim"]
    if r > 1:
        ys_in, olens_in = ys[:, r - 1::r], torch.div(olens, r, rounding_mode="floor")
    else:
        ys_in, olens_in = ys, olens
    ys_in = torch.cat([ys_in.new_zeros(ys_in.shape[0], 1, ys_in.shape[2]), ys_in[:, :-1]], dim=1)
    y_masks = N.non_pad_mask(olens_in)[:, None, :] & N.causal_mask(int(olens_in.max()))[None]
    zs = N.transformer_decoder(P(sd, "decoder."), ys_in, y_masks, hs, hs_masks, c, rt)
    before = N.linear(P(sd, "feat_out."), zs).reshape(zs.shape[0], -1, odim)
    logits = N.linear(P(sd, "prob_out."), zs).reshape(zs.shape[0], -1)
    after = before + N.postnet(P(sd, "postnet."), before.transpose(1, 2), rt).transpose(1, 2)
    if r > 1:
        olens = olens - olens % r
        mx = int(olens.max())
        ys, labels = ys[:, :mx], labels[:, :mx]
        labels = torch.scatter(labels, 1, (olens - 1).unsqueeze(1), 1.0)
    return after, before, logits, ys, labels, olens, olens_in


def vtn_forward(sd, c, xs, ilens, ys, labels, olens, training=True, drop=False):
    rt = Runtime(training, drop)
    xs = xs[:, : int(ilens.max())]
    ys, labels = ys[:, : int(olens.max())], labels[:, : int(olens.max())]
    x_masks = N.non_pad_mask(ilens)[:, None, :]
    hs, hs_masks = _vtn_encoder(sd, c, xs, x_masks, rt)
    after, before, logits, ys, labels, olens_new, olens_in = _ar_teacher_forced(sd, c, hs, hs_masks, ys, labels, olens, rt)
    ilens_ds = ((ilens - 2 + 1) // 2 - 2 + 1) // 2
    nl = N.n_layers(P(sd, "decoder."), "decoders")
    att_ws = [rt.attn[f"decoder.decoders.{i}.src_attn"] for i in reversed(range(nl))]
    return after, before, logits, ys, labels, olens_new, (att_ws, ilens_ds, olens_in), rt


def _decoder_step_all(sd, c, ys, hs, rt):
    """One `forward_one_step` of the reference recomputed from scratch over the whole prefix
    (decoder.py:239-273 + decoder_layer.py:85-132): with the per-layer output cache the result for
    the last position equals a full causal pass, which is what this restates."""
    L = ys.shape[1]
    mask = N.causal_mask(L)[None]
    zs = N.transformer_decoder(P(sd, "decoder."), ys, mask, hs, None, c, rt)
    return zs[:, -1]


def vtn_inference(sd, c, x, threshold=0.5, minlenratio=0.0, maxlenratio=10.0, drop=False, tts=False):
    """vtn.py:302-394 / transformer_tts.py:231-326; x: (T, idim) float or (T,) long for TTS."""
    rt = Runtime(False, drop)
    r, odim = _cfg(c, "decoder_reduction_factor", 2), c["odim"]
    if tts:
        x = F.pad(x, [0, 1], "constant", c["idim"] - 1)
        hs, _ = N.transformer_encoder(P(sd, "encoder."), x[None], None, dict(c, enc_dropout=0.1, enc_pos_dropout=0.1), rt, "embed-scaled")
    else:
        hs, _ = _vtn_encoder(sd, c, x[None], None, rt)
    maxlen, minlen = int(hs.shape[1] * maxlenratio / r), int(hs.shape[1] * minlenratio / r)
    ys = hs.new_zeros(1, 1, odim)
    outs, probs, atts = [], [], []
    nl = N.n_layers(P(sd, "decoder."), "decoders")
    idx = 0
    while True:
        idx += 1
        z = _decoder_step_all(sd, c, ys, hs, rt)
        outs.append(N.linear(P(sd, "feat_out."), z).view(r, odim))
        probs.append(torch.sigmoid(N.linear(P(sd, "prob_out."), z))[0])
        ys = torch.cat([ys, outs[-1][-1].view(1, 1, odim)], dim=1)
        atts.append(torch.stack([rt.attn[f"decoder.decoders.{i}.src_attn"][0, :, -1] for i in range(nl)]))  # (layers,H,T)
        if int((probs[-1] >= threshold).sum()) > 0 or idx >= maxlen:
            if idx < minlen:
                continue
            o = torch.cat(outs, dim=0)[None].transpose(1, 2)
            o = o + N.postnet(P(sd, "postnet."), o, rt)
            return o.transpose(2, 1)[0], torch.cat(probs, dim=0), torch.stack(atts, dim=2)


# =============================================================================================
# Transformer-TTS (models/transformer_tts.py:129-229)
# =============================================================================================
def tts_forward(sd, c, xs, ilens, ys, labels, olens, training=True, drop=False):
    rt = Runtime(training, drop)
    xs = xs[:, : int(ilens.max())]
    ys, labels = ys[:, : int(olens.max())], labels[:, : int(olens.max())]
    xs = F.pad(xs, [0, 1], "constant", 0)
    xs[torch.arange(xs.shape[0]), ilens] = c["idim"] - 1  # eos
    ilens = ilens + 1
    x_masks = N.non_pad_mask(ilens)[:, None, :]
    hs, hs_masks = N.transformer_encoder(P(sd, "encoder."), xs, x_masks, dict(c, enc_dropout=0.1, enc_pos_dropout=0.1), rt, "embed-scaled")
    after, before, logits, ys, labels, olens_new, olens_in = _ar_teacher_forced(sd, c, hs, hs_masks, ys, labels, olens, rt)
    return after, before, logits, ys, labels, olens_new, ([], ilens, olens_in), rt


# =============================================================================================
# AAS-VC pieces
# =============================================================================================
def alignment_module(p, text, feats, x_pad_mask):  # modules/alignments.py:28-60
    t = F.relu(F.conv1d(text.transpose(1, 2), p["t_conv1.weight"], p["t_conv1.bias"], padding=1))
    t = F.conv1d(t, p["t_conv2.weight"], p["t_conv2.bias"]).transpose(1, 2)
    f = F.relu(F.conv1d(feats.transpose(1, 2), p["f_conv1.weight"], p["f_conv1.bias"], padding=1))
    f = F.relu(F.conv1d(f, p["f_conv2.weight"], p["f_conv2.bias"], padding=1))
    f = F.conv1d(f, p["f_conv3.weight"], p["f_conv3.bias"]).transpose(1, 2)
    dist = torch.norm(f.unsqueeze(2) - t.unsqueeze(1), p=2, dim=3)
    score = -dist
    if x_pad_mask is not None:
        score = score.masked_fill(x_pad_mask.unsqueeze(-2), -np.inf)
    return F.log_softmax(score, dim=-1)


def viterbi_decode(log_p_attn, text_lens, feat_lens):  # modules/alignments.py:281-310 (differentiable bin_loss)
    ds_np, _, paths, margin = _viterbi_np(log_p_attn.detach().float().numpy(), text_lens.numpy(), feat_lens.numpy())
    B = log_p_attn.shape[0]
    bin_loss = 0
    for b in range(B):
        cur = log_p_attn[b, : int(feat_lens[b]), : int(text_lens[b])]
        bin_loss = bin_loss - cur[torch.arange(int(feat_lens[b])), torch.from_numpy(paths[b])].mean()
    return torch.from_numpy(ds_np), bin_loss / B, paths, margin


def gaussian_upsampling(hs, ds, h_masks, d_masks, delta=0.1):  # modules/length_regulator.py:111-154
    B = ds.shape[0]
    if ds.sum() == 0:
        ds = ds.clone()
        ds[ds.sum(dim=1).eq(0)] = 1
    T_feats = int(ds.sum()) if h_masks is None else h_masks.shape[-1]
    t = torch.arange(0, T_feats).unsqueeze(0).repeat(B, 1).float()
    if h_masks is not None:
        t = t * h_masks.float()
    cpos = ds.cumsum(dim=-1) - ds / 2
    energy = -1 * delta * (t.unsqueeze(-1) - cpos.unsqueeze(1)) ** 2
    if d_masks is not None:
        energy = energy.masked_fill(~(d_masks.unsqueeze(1).repeat(1, T_feats, 1)), -float("inf"))
    return torch.matmul(torch.softmax(energy, dim=2).to(hs.dtype), hs)       # (.to: no-op in the reference's fp32)


# --------------------------------------------------------------------------- VITS flows (modules/vits/*.py)
def _dds_conv(p, x, x_mask, rt, drop_p, g=None, eps=1e-5):  # vits/flow.py:110-190
    if g is not None:
        x = x + g
    i = 0
    while p.has(f"convs.{i}.0.weight"):
        w = p[f"convs.{i}.0.weight"]
        k = w.shape[-1]
        dil = k ** i
        y = F.conv1d(x * x_mask, w, p[f"convs.{i}.0.bias"], padding=(k * dil - dil) // 2, dilation=dil, groups=w.shape[0])
        y = F.gelu(F.layer_norm(y.transpose(1, 2), (y.shape[1],), p[f"convs.{i}.2.weight"], p[f"convs.{i}.2.bias"], eps).transpose(1, 2))
        y = F.conv1d(y, p[f"convs.{i}.5.weight"], p[f"convs.{i}.5.bias"])
        y = F.gelu(F.layer_norm(y.transpose(1, 2), (y.shape[1],), p[f"convs.{i}.7.weight"], p[f"convs.{i}.7.bias"], eps).transpose(1, 2))
        x = x + rt.dropout(y, drop_p)
        i += 1
    return x * x_mask


def _rq_spline(x, uw, uh, ud, inverse, bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """Piecewise rational-quadratic spline with linear tails (vits/transform.py:44-216), evaluated for
    every element with torch.where instead of boolean-mask gathers."""
    nb = uw.shape[-1]
    inside = (x >= -bound) & (x <= bound)
    const = math.log(math.exp(1 - min_d) - 1)
    ud = F.pad(ud, (1, 1))
    ud[..., 0] = const
    ud[..., -1] = const
    xin = torch.where(inside, x, torch.zeros_like(x))

    def knots(u, lo, hi, mn):
        s = mn + (1 - mn * nb) * F.softmax(u, dim=-1)
        cs = F.pad(torch.cumsum(s, dim=-1), (1, 0), value=0.0)
        cs = (hi - lo) * cs + lo
        cs[..., 0] = lo
        cs[..., -1] = hi
        return cs, cs[..., 1:] - cs[..., :-1]

    cw, w = knots(uw, -bound, bound, min_w)
    ch, h = knots(uh, -bound, bound, min_h)
    d = min_d + F.softplus(ud)
    loc = (ch if inverse else cw).clone()
    loc[..., -1] += 1e-6
    idx = (torch.sum(xin[..., None] >= loc, dim=-1) - 1)[..., None]
    g = lambda a: a.gather(-1, idx)[..., 0]
    in_cw, in_w, in_ch, in_h = g(cw), g(w), g(ch), g(h)
    delta = h / w
    in_delta, in_d, in_d1 = g(delta), g(d), g(d[..., 1:])
    if inverse:
        a = (xin - in_ch) * (in_d + in_d1 - 2 * in_delta) + in_h * (in_delta - in_d)
        b = in_h * in_d - (xin - in_ch) * (in_d + in_d1 - 2 * in_delta)
        cq = -in_delta * (xin - in_ch)
        root = (2 * cq) / (-b - torch.sqrt(b.pow(2) - 4 * a * cq))
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


def _conv_flow(p, x, x_mask, g, rt, inverse=False, bins=10):  # vits/flow.py:250-310
    xa, xb = x.split(x.shape[1] // 2, 1)
    hidden = p["input_conv.weight"].shape[0]
    hh = F.conv1d(xa, p["input_conv.weight"], p["input_conv.bias"])
    hh = _dds_conv(p.sub("dds_conv"), hh, x_mask, rt, 0.0, g=g)
    hh = F.conv1d(hh, p["proj.weight"], p["proj.bias"]) * x_mask
    b, c, t = xa.shape
    hh = hh.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    den = math.sqrt(hidden)
    xb, lad = _rq_spline(xb, hh[..., :bins] / den, hh[..., bins:2 * bins] / den, hh[..., 2 * bins:], inverse)
    y = torch.cat([xa, xb], 1) * x_mask
    return y, torch.sum(lad * x_mask, [1, 2])


def _affine_flow(p, x, x_mask, inverse=False):  # vits/flow.py:66-93
    if not inverse:
        return (p["m"] + torch.exp(p["logs"]) * x) * x_mask, torch.sum(p["logs"] * x_mask, [1, 2])
    return (x - p["m"]) * torch.exp(-p["logs"]) * x_mask, None


def _n_flows(p, stem):
    n = 0
    while p.has(f"{stem}.{2 * n + 1}.input_conv.weight"):
        n += 1
    return n


def sdp_forward(p, x, x_mask, w, noise, rt, drop_p=0.5):
    """StochasticDurationPredictor NLL (modules/duration_predictor.py:211-280); `noise` replaces the
    reference's torch.randn(B, 2, T) draw."""
    x = F.conv1d(x.detach(), p["pre.weight"], p["pre.bias"])
    x = _dds_conv(p.sub("dds"), x, x_mask, rt, drop_p)
    x = F.conv1d(x, p["proj.weight"], p["proj.bias"]) * x_mask
    h_w = F.conv1d(w, p["post_pre.weight"], p["post_pre.bias"])
    h_w = _dds_conv(p.sub("post_dds"), h_w, x_mask, rt, drop_p)
    h_w = F.conv1d(h_w, p["post_proj.weight"], p["post_proj.bias"]) * x_mask
    e_q = noise * x_mask
    z_q, ld = _affine_flow(p.sub("post_flows.0"), e_q, x_mask)
    logdet_q = ld
    for i in range(_n_flows(p, "post_flows")):
        z_q, ld = _conv_flow(p.sub(f"post_flows.{2 * i + 1}"), z_q, x_mask, x + h_w, rt)
        logdet_q = logdet_q + ld
        z_q = torch.flip(z_q, [1])
    z_u, z1 = torch.split(z_q, [1, 1], 1)
    u = torch.sigmoid(z_u) * x_mask
    z0 = (w - u) * x_mask
    logdet_q = logdet_q + torch.sum((F.logsigmoid(z_u) + F.logsigmoid(-z_u)) * x_mask, [1, 2])
    logq = torch.sum(-0.5 * (math.log(2 * math.pi) + (e_q ** 2)) * x_mask, [1, 2]) - logdet_q
    z0 = torch.log(torch.clamp_min(z0, 1e-5)) * x_mask      # LogFlow, vits/flow.py:37-63
    logdet = torch.sum(-z0, [1, 2])
    z = torch.cat([z0, z1], 1)
    z, ld = _affine_flow(p.sub("flows.0"), z, x_mask)
    logdet = logdet + ld
    for i in range(_n_flows(p, "flows")):
        z, ld = _conv_flow(p.sub(f"flows.{2 * i + 1}"), z, x_mask, x, rt)
        logdet = logdet + ld
        z = torch.flip(z, [1])
    nll = torch.sum(0.5 * (math.log(2 * math.pi) + (z ** 2)) * x_mask, [1, 2]) - logdet
    return nll + logq


def sdp_inverse(p, x, x_mask, noise, rt, noise_scale=0.8):
    """duration_predictor.py:281-304: flows reversed, first ConvFlow ('useless vflow') dropped."""
    x = F.conv1d(x.detach(), p["pre.weight"], p["pre.bias"])
    x = _dds_conv(p.sub("dds"), x, x_mask, rt, 0.5)
    x = F.conv1d(x, p["proj.weight"], p["proj.bias"]) * x_mask
    nf = _n_flows(p, "flows")
    # module list: [affine, conv1, flip, conv2, flip, ..., conv_nf, flip]; reversed minus [-2] (conv1)
    seq = ["affine"]
    for i in range(nf):
        seq += [("conv", 2 * i + 1), "flip"]
    seq = list(reversed(seq))
    seq = seq[:-2] + [seq[-1]]
    z = noise * noise_scale
    for s in seq:
        if s == "flip":
            z = torch.flip(z, [1])
        elif s == "affine":
            z, _ = _affine_flow(p.sub("flows.0"), z, x_mask, inverse=True)
        else:
            z, _ = _conv_flow(p.sub(f"flows.{s[1]}"), z, x_mask, x, rt, inverse=True)
    z0, _ = z.split(1, 1)
    return torch.ceil(torch.exp(z0) * x_mask)


def duration_predictor(p, xs, x_masks, rt, drop_p, inference=False):  # modules/duration_predictor.py:81-100
    x = xs.transpose(1, -1)
    i = 0
    while p.has(f"conv.{i}.0.weight"):
        w = p[f"conv.{i}.0.weight"]
        x = F.relu(F.conv1d(x, w, p[f"conv.{i}.0.bias"], padding=(w.shape[-1] - 1) // 2))
        x = F.layer_norm(x.transpose(1, -1), (x.shape[1],), p[f"conv.{i}.2.weight"], p[f"conv.{i}.2.bias"], 1e-12).transpose(1, -1)
        x = rt.dropout(x, drop_p)
        i += 1
    x = N.linear(p.sub("linear"), x.transpose(1, -1)).squeeze(-1)
    if inference:
        x = torch.clamp(torch.round(x.exp() - 1.0), min=0).long()
    if x_masks is not None:
        x = x * x_masks
    return x


# =============================================================================================
# AAS-VC (models/aas_vc.py:279-529)
# =============================================================================================
MAX_DP_OUTPUT = 10


def aasvc_forward(sd, c, xs, ilens, ys, olens, dp_inputs=None, noise=None, training=True, drop=False, inference=False):
    rt = Runtime(training, drop)
    adim, heads = _cfg(c, "adim", 384), _cfg(c, "aheads", 4)
    er, pr, dr = _cfg(c, "encoder_reduction_factor", 1), _cfg(c, "post_encoder_reduction_factor", 1), _cfg(c, "decoder_reduction_factor", 1)
    odim = c["odim"]
    if not inference:
        xs = xs[:, : int(ilens.max())]
        ys = ys[:, : int(olens.max())]
    ret = {}
    if er > 1:
        b, tmax, dim = xs.shape
        if tmax % er:
            xs = xs[:, : -(tmax % er)]
        xs = xs.reshape(b, tmax // er, dim * er)
        ilens = ilens // er
    inl = c.get("encoder_input_layer", "linear")
    pos_type = c.get("conformer_pos_enc_layer_type", "rel_pos")
    ff = c.get("positionwise_layer_type", "conv1d")
    hs, _ = N.conformer_encoder(P(sd, "encoder."), xs, N.non_pad_mask(ilens)[:, None, :], c, rt, heads, inl, pos_type,
                                _cfg(c, "transformer_enc_dropout_rate", 0.1), _cfg(c, "transformer_enc_positional_dropout_rate", 0.1),
                                _cfg(c, "transformer_enc_attn_dropout_rate", 0.1), c.get("encoder_normalize_before", False),
                                "encoder", ff)
    if inl == "conv2d":
        ilens = ((ilens - 2 + 1) // 2 - 2 + 1) // 2
    if pr > 1:
        b, tmax, dim = hs.shape
        if tmax % pr:
            hs = hs[:, : -(tmax % pr)]
        hs = hs.reshape(b, tmax // pr, dim * pr)
        ilens = ilens // pr
    if c.get("duration_predictor_use_encoder_outputs", True):
        dpi = hs
    else:
        dpi, _ = N.conv2d_subsample(P(sd, "duration_predictor_projection."), dp_inputs, None)
        dpi = torch.stack([F.interpolate(dpi[i][None].permute(0, 2, 1), size=hs.shape[1]).permute(0, 2, 1)[0]
                           for i in range(dpi.shape[0])])
    olens_red = olens
    if dr > 1 and ys is not None:
        b, tmax, dim = ys.shape
        if tmax % dr:
            ys = ys[:, : -(tmax % dr)]
        ys = ys.reshape(b, tmax // dr, dim * dr)
        olens_red = olens // dr
    h_pad = N.pad_mask(ilens)
    h_np = N.non_pad_mask(ilens)
    stochastic = c.get("duration_predictor_type", "deterministic") == "stochastic"
    dp = P(sd, "duration_predictor.")
    if inference:
        log_p_attn, ds, bin_loss = None, None, 0.0
        if ys is not None:
            log_p_attn = alignment_module(P(sd, "alignment_module."), hs, ys, h_pad)
            ds, bin_loss, _, _ = viterbi_decode(log_p_attn, ilens, olens_red)
        if stochastic:
            d_outs = sdp_inverse(dp, dpi.transpose(1, 2), h_np.unsqueeze(1).float(), noise, rt,
                                 _cfg(c, "stochastic_duration_predictor_noise_scale", 0.8)).squeeze(1)
        else:
            d_outs = duration_predictor(dp, dpi, None, rt, _cfg(c, "duration_predictor_dropout_rate", 0.1), inference=True)
        d_outs = torch.clamp(d_outs, max=MAX_DP_OUTPUT)
        ret["d_outs"] = d_outs
        hs = gaussian_upsampling(hs, d_outs.float(), None, h_np)
    else:
        log_p_attn = alignment_module(P(sd, "alignment_module."), hs, ys, h_pad)
        ds, bin_loss, paths, margin = viterbi_decode(log_p_attn, ilens, olens_red)
        ret["mas_paths"], ret["mas_margin"] = paths, margin
        if stochastic:
            # (.to(dpi.dtype): fp32 in the reference; lets the tests evaluate this restatement in float64 as their yardstick)
            nll = sdp_forward(dp, dpi.transpose(1, 2), h_np.unsqueeze(1).to(dpi.dtype), ds.unsqueeze(1).to(dpi.dtype), noise, rt,
                              _cfg(c, "stochastic_duration_predictor_dropout_rate", 0.5))
            ret["dur_nll"] = nll / torch.sum(h_np)
        else:
            d_outs = duration_predictor(dp, dpi, h_np, rt, _cfg(c, "duration_predictor_dropout_rate", 0.1))
            ret["d_outs"] = torch.clamp(d_outs, max=MAX_DP_OUTPUT)
        hs = gaussian_upsampling(hs, ds, N.non_pad_mask(olens_red), h_np)
    h_masks = N.non_pad_mask(olens_red)[:, None, :] if (olens is not None and not inference) else None
    zs, _ = N.conformer_encoder(P(sd, "decoder."), hs, h_masks, c, rt, heads, None, pos_type,
                                _cfg(c, "transformer_dec_dropout_rate", 0.1), _cfg(c, "transformer_dec_positional_dropout_rate", 0.1),
                                _cfg(c, "transformer_dec_attn_dropout_rate", 0.1), c.get("decoder_normalize_before", False),
                                "decoder", ff)
    before = N.linear(P(sd, "feat_out."), zs).reshape(zs.shape[0], -1, odim)
    if P(sd, "postnet.").has("postnet.0.0.weight"):
        after = before + N.postnet(P(sd, "postnet."), before.transpose(1, 2), rt, _cfg(c, "postnet_dropout_rate", 0.5)).transpose(1, 2)
    else:
        after = before
    ret.update(before_outs=before, after_outs=after, ds=ds, ilens=ilens, bin_loss=bin_loss, log_p_attn=log_p_attn,
               olens_reduced=olens_red)
    if not inference:
        if dr > 1:
            olens = olens - olens % dr
            ys = ys[:, : int(olens.max())]
        ret["olens"], ret["ys"] = olens, ys
    ret["_rt"] = rt
    return ret


# =============================================================================================
# FastSpeechVC (models/fastspeech_vc.py:244-466), LengthRegulator (modules/length_regulator.py:46-97),
# DurationCalculator (utils/duration_calculator.py:13-65)
# =============================================================================================
def length_regulator(xs, ds, alpha=1.0, pad_value=0.0):
    if alpha != 1.0:
        ds = torch.round(ds.float() * alpha).long()
    if ds.sum() == 0:
        ds = ds.clone()
        ds[ds.sum(dim=1).eq(0)] = 1
    rep = [torch.repeat_interleave(x, d, dim=0) for x, d in zip(xs, ds)]
    tmax = max(r.shape[0] for r in rep)
    out = xs.new_full((len(rep), tmax) + tuple(xs.shape[2:]), pad_value)
    for i, r in enumerate(rep):
        out[i, : r.shape[0]] = r
    return out


def duration_calculator(att_ws):
    if att_ws.dim() == 4:
        a = att_ws.reshape(-1, att_ws.shape[-2], att_ws.shape[-1])
        scores = a.max(dim=-1)[0].mean(dim=-1)
        focus = scores.max()
        a = a[scores.argmax()]
    else:
        a = att_ws
        focus = a.max(dim=-1)[0].mean()
    durations = torch.stack([a.argmax(-1).eq(i).sum() for i in range(a.shape[1])])
    return durations.view(-1), focus


def fastspeech_vc_forward(sd, c, xs, ilens, ys=None, olens=None, ds=None, dp_inputs=None, training=True, drop=False,
                          inference=False, alpha=1.0):
    rt = Runtime(training, drop)
    adim, heads = _cfg(c, "adim", 384), _cfg(c, "aheads", 4)
    er, dr = _cfg(c, "encoder_reduction_factor", 1), _cfg(c, "decoder_reduction_factor", 1)
    teacher_r = _cfg(c, "teacher_model_decoder_reduction_factor", 4)
    odim = c["odim"]
    if not inference:
        xs = xs[:, : int(ilens.max())]
        ys = ys[:, : int(olens.max())]
    if er > 1:
        b, tmax, dim = xs.shape
        if tmax % er:
            xs = xs[:, : -(tmax % er)]
        xs = xs.reshape(b, tmax // er, dim * er)
        ilens = ilens // er
    enc_type = c.get("encoder_type", "transformer")
    inl = c.get("encoder_input_layer", "linear")
    pos_type = c.get("conformer_pos_enc_layer_type", "rel_pos")
    ff = c.get("positionwise_layer_type", "conv1d")
    x_masks = N.non_pad_mask(ilens)[:, None, :]
    if enc_type == "transformer":
        hs, _ = _vtn_encoder(sd, dict(c, encoder_type="transformer"), xs, x_masks, rt)
    else:
        hs, _ = N.conformer_encoder(P(sd, "encoder."), xs, x_masks, c, rt, heads, inl, pos_type,
                                    _cfg(c, "transformer_enc_dropout_rate", 0.1), _cfg(c, "transformer_enc_positional_dropout_rate", 0.1),
                                    _cfg(c, "transformer_enc_attn_dropout_rate", 0.1), c.get("encoder_normalize_before", False),
                                    "encoder", ff)
    if inl == "conv2d":
        ilens = ((ilens - 2 + 1) // 2 - 2 + 1) // 2
    if c.get("duration_predictor_use_encoder_outputs", True):
        dpi = hs
    else:
        dpi, _ = N.conv2d_subsample(P(sd, "duration_predictor_projection."), dp_inputs, None)
        dpi = torch.stack([F.interpolate(dpi[i][None].permute(0, 2, 1), size=hs.shape[1]).permute(0, 2, 1)[0]
                           for i in range(dpi.shape[0])])
    dp = P(sd, "duration_predictor.")
    if inference:
        d_outs = duration_predictor(dp, dpi, None, rt, _cfg(c, "duration_predictor_dropout_rate", 0.1), inference=True)
        hs = length_regulator(hs, d_outs * teacher_r, alpha)
        h_masks = None
    else:
        d_outs = duration_predictor(dp, dpi, N.non_pad_mask(ilens, dpi.shape[1]), rt, _cfg(c, "duration_predictor_dropout_rate", 0.1))
        hs = length_regulator(hs, ds * teacher_r)
        olens_in = olens // dr if dr > 1 else olens
        h_masks = N.non_pad_mask(olens_in)[:, None, :]
    zs, _ = N.conformer_encoder(P(sd, "decoder."), hs, h_masks, c, rt, heads, None, pos_type,
                                _cfg(c, "transformer_dec_dropout_rate", 0.1), _cfg(c, "transformer_dec_positional_dropout_rate", 0.1),
                                _cfg(c, "transformer_dec_attn_dropout_rate", 0.1), c.get("decoder_normalize_before", False),
                                "decoder", ff)
    before = N.linear(P(sd, "feat_out."), zs).reshape(zs.shape[0], -1, odim)
    if P(sd, "postnet.").has("postnet.0.0.weight"):
        after = before + N.postnet(P(sd, "postnet."), before.transpose(1, 2), rt, _cfg(c, "postnet_dropout_rate", 0.5)).transpose(1, 2)
    else:
        after = before
    if inference:
        return before, after, d_outs, ilens
    if dr > 1:
        olens = olens - olens % dr
        ys = ys[:, : int(olens.max())]
    return before, after, d_outs, ilens, olens, ys


# =============================================================================================
# losses (losses/*.py)
# =============================================================================================
def seq2seq_loss(after, before, logits, ys, labels, olens, bce_pos_weight=10.0):  # losses/seq2seq_loss.py:30-59
    m = N.non_pad_mask(olens, ys.shape[1]).unsqueeze(-1)
    ys_, a_, b_ = ys.masked_select(m), after.masked_select(m), before.masked_select(m)
    lab, lg = labels.masked_select(m[:, :, 0]), logits.masked_select(m[:, :, 0])
    l1 = F.l1_loss(a_, ys_) + F.l1_loss(b_, ys_)
    bce = F.binary_cross_entropy_with_logits(lg, lab, pos_weight=torch.tensor(bce_pos_weight))
    return l1, bce


def l1_loss(after, before, ys, olens):  # losses/l1_loss.py:22-49
    m = N.non_pad_mask(olens, ys.shape[1]).unsqueeze(-1)
    return F.l1_loss(before.masked_select(m), ys.masked_select(m)) + F.l1_loss(after.masked_select(m), ys.masked_select(m))


def betabinom_logprior(T, Nn, w=1.0):
    """log pmf of BetaBinomial(n=N, a=w*t, b=w*(T-t+1)) at k=0..N-1 for t=1..T  -> (T, N) float64.
    Closed form of scipy.stats.betabinom.logpmf used at losses/forward_sum_loss.py:100-107."""
    from scipy.special import betaln, gammaln
    t = np.arange(1, T + 1, dtype=float)[None, :]
    a, b = w * t, w * (T - t + 1)
    k = np.arange(Nn, dtype=float)[:, None]
    comb = gammaln(Nn + 1) - gammaln(k + 1) - gammaln(Nn - k + 1)
    return (comb + betaln(k + a, Nn - k + b) - betaln(a, b)).T  # (T, N)


def forward_sum_loss(log_p_attn, ilens, olens, blank_prob=np.e ** -1):  # losses/forward_sum_loss.py:26-76
    B = log_p_attn.shape[0]
    T_text, T_feats = int(ilens.max()), int(olens.max())
    prior = torch.full((B, T_feats, T_text), -np.inf)
    for b in range(B):
        T, Nn = int(olens[b]), int(ilens[b])
        prior[b, :T, :Nn] = torch.from_numpy(betabinom_logprior(T, Nn))
    lp = log_p_attn + prior.to(log_p_attn.dtype)
    lp = F.pad(lp, (1, 0, 0, 0, 0, 0), value=float(np.log(blank_prob)))
    loss = 0
    for b in range(B):
        tgt = torch.arange(1, int(ilens[b]) + 1).unsqueeze(0)
        cur = lp[b, : int(olens[b]), : int(ilens[b]) + 1].unsqueeze(1)
        loss = loss + F.ctc_loss(cur, tgt, input_lengths=olens[b:b + 1], target_lengths=ilens[b:b + 1], zero_infinity=True)
    return loss / B


def guided_attention_loss(att_ws, ilens, olens, sigma=0.4, alpha=1.0):  # losses/guided_attention_loss.py:142-165
    B, _, To, Ti = att_ws.shape
    w = torch.zeros(B, To, Ti)
    for b in range(B):
        il, ol = int(ilens[b]), int(olens[b])
        gx, gy = torch.meshgrid(torch.arange(ol), torch.arange(il), indexing="ij")
        w[b, :ol, :il] = 1.0 - torch.exp(-((gy.float() / il - gx.float() / ol) ** 2) / (2 * sigma ** 2))
    m = (N.non_pad_mask(olens, To).unsqueeze(-1) & N.non_pad_mask(ilens, Ti).unsqueeze(-2)).unsqueeze(1)
    return alpha * torch.mean((w.unsqueeze(1) * att_ws).masked_select(m))


def duration_predictor_loss(d_outs, ds, ilens, offset=1.0):  # losses/duration_predictor_loss.py:38-57
    m = N.non_pad_mask(ilens, ds.shape[1])
    return F.mse_loss(d_outs.masked_select(m), torch.log(ds.masked_select(m).float() + offset))


# =============================================================================================
# optimiser-step replays (trainers/ar_vc.py:59-112, trainers/aas_vc.py:56-164, schedulers/warmup_lr.py:54-61)
# =============================================================================================
def warmup_lr(base_lr, step_num, warmup_steps=4000):
    return base_lr * warmup_steps ** 0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


def adam_step(params, grads, state, lr, step, betas=(0.9, 0.999), eps=1e-8, grad_clip=1.0):
    """clip_grad_norm_(max_norm) -> torch.optim.Adam update (no weight decay, no amsgrad); in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(grad_clip / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for p, g, (m, v) in zip(params, grads, state):
        g = g * coef
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
    return total
