"""Transformer layers as hand-scheduled kernel sequences (bf16 training path): ONE autograd node per layer, whose forward and
backward passes are a few fused launches each instead of one node -- and one or more launch-bound kernels -- per operation.

    encoder layer, pre-LN  (modules/transformer/encoder_layer.py:61-119, normalize_before=True):
        forward  4 launches (was 7):  [LN1 + Q|K|V projection + attention]  [out-proj + dropout + residual]
                                      [LN2 + w_1 + ReLU + dropout]           [w_2 + dropout + residual]
        backward 5 launches (was 7):  [dropmask + dgrad w_2 * relu' * dropmask]  [dgrad w_1]
                                      [LN2' + dropmask + dgrad out-proj + attention']  [dgrad Q|K|V]  [LN1']
    decoder layer, post-LN (modules/transformer/decoder_layer.py:63-134, normalize_before=False):
        forward  6 launches (was 11), backward 7 (was 11); the LayerNorm that ends a layer is the prologue of the next
        layer's first kernel (the layer functions exchange the PRE-norm stream), the last one runs on its own.

The kernels: csrc/attn_block.hip (attention sub-layer: one workgroup per (utterance, head)), csrc/gemm_rowpro.hip (GEMM with
LayerNorm / dropout-mask row prologue), the LDS-DMA GEMMs with dropout + residual in their epilogue, norm.hip's LayerNorm
backward.  Parameter gradients go to the flat-gradient slots of optim.FlatAdam on the side streams exactly as in
ops/functional.py (weight-gradient GEMMs with fused bias row sums, grouped column reductions for the LayerNorm vectors).

Eligibility (`enc_layer_ok` / `dec_layer_ok`): compute dtype bf16, T <= 64, (D, d_k) in {(256, 64), (384, 96)},
ReLU feed-forward, every parameter of the layer trainable and held by FlatAdam with bf16 + transposed shadows.  Everything
else (fp32 parity mode, frozen layers, long sequences, Conformer) takes the modular path of modules.py, which is also the
reference these functions are tested against (tests/gpu_kernel_check.py: fused_layers_vs_modular).
"""
import torch
from torch.autograd import Function

from . import functional as Fn
from . import kernels as K
from . import kernels_block as KB

_NOSEED = (None, 0)


def _seed(p, dev):
    return K.new_seed(dev) if p > 0.0 else _NOSEED


def _w(t):
    return t._s2s_bf16


def _wt(t):
    t._s2s_perm_registry.sync()            # refreshed by the step prologue (optim.FlatAdam.begin_step)
    return t._s2s_bf16_t


def _has_shadows(*ws):
    return all(getattr(w, "_s2s_bf16", None) is not None and getattr(w, "_s2s_bf16_t", None) is not None and
               getattr(w, "_s2s_grad", None) is not None for w in ws)


def _slotted(*ps):
    return all(p is not None and p.requires_grad and getattr(p, "_s2s_grad", None) is not None for p in ps)


def _attn_ok(att, need_qkv):
    f = getattr(att, "_fused", None)
    if f is None or ("w_qkv" if need_qkv else "w_q") not in f:
        return False
    ws = [f["w_qkv"]] if need_qkv else [f["w_q"]]
    bs = [f["b_qkv"]] if need_qkv else [f["b_q"]]
    return _has_shadows(att.linear_out.weight, *ws) and _slotted(att.linear_out.bias, *bs)


def _ffn_ok(ff):
    from ..modules import PositionwiseFeedForward
    return (type(ff) is PositionwiseFeedForward and ff.activation == "relu" and _has_shadows(ff.w_1.weight, ff.w_2.weight) and
            _slotted(ff.w_1.bias, ff.w_2.bias) and ff.w_1.weight.shape[0] % 8 == 0)


def _norm_ok(*norms):
    return all(_slotted(n.weight, n.bias) and n.weight.data_ptr() % 16 == 0 and n.bias.data_ptr() % 16 == 0 for n in norms)


def enc_layer_ok(layer, x):
    from ..modules import MultiHeadedAttention
    if not (x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 3 and layer.normalize_before):
        return False
    B, T, D = x.shape
    att = layer.self_attn
    return (type(att) is MultiHeadedAttention and KB.block_supported(x.dtype, T, T, D, att.h) and KB.rowpro_supported(x.dtype, D) and
            _attn_ok(att, True) and _ffn_ok(layer.feed_forward) and _norm_ok(layer.norm1, layer.norm2))


def dec_self_ok(layer, x):
    from ..modules import MultiHeadedAttention
    if not (x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 3 and not layer.normalize_before):
        return False
    B, T, D = x.shape
    sa = layer.self_attn
    return type(sa) is MultiHeadedAttention and KB.block_supported(x.dtype, T, T, D, sa.h) and _attn_ok(sa, True) and _norm_ok(layer.norm1)


def dec_layer_ok(layer, x, memory):
    from ..modules import MultiHeadedAttention
    if not (x.dtype == torch.bfloat16 and x.is_cuda and x.dim() == 3 and not layer.normalize_before and memory is not None):
        return False
    B, T, D = x.shape
    sa, ca = layer.self_attn, layer.src_attn
    return (type(sa) is MultiHeadedAttention and type(ca) is MultiHeadedAttention and sa.h == ca.h and
            KB.block_supported(x.dtype, T, T, D, sa.h) and KB.block_supported(x.dtype, T, memory.shape[1], D, ca.h) and
            KB.rowpro_supported(x.dtype, D) and _attn_ok(sa, True) and _attn_ok(ca, False) and _ffn_ok(layer.feed_forward) and
            _norm_ok(layer.norm1, layer.norm2, layer.norm3))


# ------------------------------------------------------------------------------------------------------------------
# shared pieces
# ------------------------------------------------------------------------------------------------------------------
def _proj_res(ctx2, lin, res2, p, seed):
    """s = res + dropout(ctx . W^T + b): ONE GEMM (dropout mask and residual in its epilogue)."""
    M, D = res2.shape
    s = torch.empty_like(res2)
    K.gemm(K.operand(ctx2, ctx2.shape[1]), K.operand(_w(lin.weight), ctx2.shape[1]), M, D, ctx2.shape[1], s, in_dtype=ctx2.dtype,
           bias=lin.bias, res=res2, drop_p=p, seed=seed)
    return s


def _wgrad(weight, bias, g2, a2):
    """dW (+)= g^T a and db (+)= column sums of g, queued for the side streams (slots of the flat gradient buffer)."""
    n_out, n_in = g2.shape[1], a2.shape[1]
    M = g2.shape[0]
    dtype = g2.dtype
    tile, sk = K.plan_gemm(n_out, n_in, M)
    rs, racc, _ = Fn._bias_sink(bias, n_out)

    def wr(out, acc):
        K.gemm(K.operand(g2, n_out, layout=K.RC), K.operand(a2, n_in, layout=K.RC), n_out, n_in, M, out, in_dtype=dtype, splitk=sk,
               tile=tile, accumulate=acc, a_rowsum=rs, a_rowsum_accumulate=racc)
    Fn._side_run(lambda: Fn._emit_wgrad(weight, (n_out, n_in), wr), keep=(g2, a2))


def _norm_grads(norm, dy2, s2, mean, rstd):
    Fn._side_run(lambda: Fn._reduce_to(norm.bias, norm.weight, 1, dy2, s2, mean, rstd), keep=(dy2, s2, mean, rstd))


def _dgrad(g2, weight, n_out, n_in, res=None):
    """dX[M, n_in] = g[M, n_out] . W[n_out, n_in] (+ res): the LDS-DMA GEMM over the transposed bf16 shadow."""
    M = g2.shape[0]
    dx = torch.empty((M, n_in), dtype=g2.dtype, device=g2.device)
    K.gemm(K.operand(g2, n_out), K.operand(_wt(weight), n_out), M, n_in, n_out, dx, in_dtype=g2.dtype, res=res)
    return dx


# ------------------------------------------------------------------------------------------------------------------
# encoder layer, pre-LN
# ------------------------------------------------------------------------------------------------------------------
class _EncLayerPreLN(Function):
    @staticmethod
    def forward(ctx, x, layer, klen, p_res, p_attn, p_ffn):
        B, T, D = x.shape
        dev = x.device
        x = Fn._c(x)
        att, ff = layer.self_attn, layer.feed_forward
        f = att._fused
        n1, n2 = layer.norm1, layer.norm2
        seed_a = _seed(p_attn, dev)
        cv, attn, qkv, y1, mean1, rstd1 = KB.attn_block_fwd(x, (n1.weight, n1.bias, n1.eps), _w(f["w_qkv"]), f["b_qkv"], att.h, klen,
                                                            False, p_attn, seed_a)
        seed_r1 = _seed(p_res, dev)
        x2 = x.view(B * T, D)
        s1 = _proj_res(cv.view(B * T, D), att.linear_out, x2, p_res, seed_r1)
        Hd = ff.w_1.weight.shape[0]
        hmid = torch.empty((B * T, Hd), dtype=x.dtype, device=dev)
        seed_h = _seed(p_ffn, dev)
        y2, mean2, rstd2 = KB.gemm_rowpro(s1, _w(ff.w_1.weight), Hd, hmid, mode=1, norm=(n2.weight, n2.bias, n2.eps), bias=ff.w_1.bias,
                                          act="relu", drop_p=p_ffn, seed=seed_h)
        seed_r2 = _seed(p_res, dev)
        out = torch.empty_like(x2)
        K.gemm(K.operand(hmid, Hd), K.operand(_w(ff.w_2.weight), Hd), B * T, D, Hd, out, in_dtype=x.dtype, bias=ff.w_2.bias, res=s1,
               drop_p=p_res, seed=seed_r2)
        ctx.layer = layer
        ctx.meta = (p_res, p_attn, p_ffn, seed_a, seed_r1, seed_h, seed_r2, (B, T, D))
        ctx.save_for_backward(x, y1, mean1, rstd1, qkv, attn, cv, s1, y2, mean2, rstd2, hmid)
        ctx.set_materialize_grads(False)
        return out.view(B, T, D), Fn._user_attn(attn, T)

    @staticmethod
    def backward(ctx, dout, dattn):
        x, y1, mean1, rstd1, qkv, attn, cv, s1, y2, mean2, rstd2, hmid = ctx.saved_tensors
        layer = ctx.layer
        p_res, p_attn, p_ffn, seed_a, seed_r1, seed_h, seed_r2, (B, T, D) = ctx.meta
        att, ff = layer.self_attn, layer.feed_forward
        f = att._fused
        n1, n2 = layer.norm1, layer.norm2
        M, Hd = B * T, hmid.shape[1]
        dev = x.device
        if dout is None:
            dout = torch.zeros((B, T, D), dtype=x.dtype, device=dev)
        g = Fn._c(dout).view(M, D)
        # feed-forward block: du = ((g * mask) W2) * relu' * mask_h ;  dy2 = du W1
        du = torch.empty((M, Hd), dtype=x.dtype, device=dev)
        daf, _, _ = KB.gemm_rowpro(g, _wt(ff.w_2.weight), Hd, du, mode=2, p_a=p_res, seed_a=seed_r2, write_rows=True, emask=hmid,
                                   drop_p=p_ffn, seed=seed_h)
        dy2 = _dgrad(du, ff.w_1.weight, Hd, D)
        # attention sub-layer: ds1 = g + LN2'(dy2) ; da = ds1 * mask ; dctx = da Wo ; attention' -> dqkv
        dqkv = torch.empty_like(qkv)
        ds1, da = KB.attn_block_bwd(dy2.view(B, T, D), (s1, mean2, rstd2, n2.weight), g, p_res, 1.0, seed_r1, _wt(att.linear_out.weight),
                                    qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], attn, Fn._pad_like(dattn, attn), att.h, p_attn, seed_a,
                                    dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:])
        dqkv2 = dqkv.view(M, 3 * D)
        dy1 = _dgrad(dqkv2, f["w_qkv"], 3 * D, D)
        dx, _ = K.layernorm_bwd(dy1, x.view(M, D), mean1, rstd1, n1.weight, ds_extra=ds1.view(M, D))
        # parameter gradients (side streams)
        _wgrad(ff.w_2.weight, ff.w_2.bias, daf, hmid)
        _wgrad(ff.w_1.weight, ff.w_1.bias, du, y2)
        _norm_grads(n2, dy2, s1, mean2, rstd2)
        _wgrad(att.linear_out.weight, att.linear_out.bias, da.view(M, D), cv.view(M, D))
        _wgrad(f["w_qkv"], f["b_qkv"], dqkv2, y1.view(M, D))
        _norm_grads(n1, dy1, x.view(M, D), mean1, rstd1)
        return dx.view(B, T, D), None, None, None, None, None


def enc_layer(layer, x, klens):
    """x_out of a pre-LN EncoderLayer; sets layer.self_attn.attn."""
    tr = layer.training
    att = layer.self_attn
    out, attn = _EncLayerPreLN.apply(x, layer, None if klens is None else klens.dev, layer.dropout_rate if tr else 0.0,
                                     att.dropout_rate if tr else 0.0, layer.feed_forward.dropout_rate if tr else 0.0)
    att.attn = attn
    return out


# ------------------------------------------------------------------------------------------------------------------
# decoder layer, post-LN.  The functions exchange the PRE-norm stream s; `prev_norm` (the norm that ends the previous layer,
# None in front of the first layer) is applied by the first kernel of the layer.
# ------------------------------------------------------------------------------------------------------------------
def _dec_self_fwd(s_in, prev_norm, layer, klen, causal, p_res, p_attn):
    B, T, D = s_in.shape
    att = layer.self_attn
    f = att._fused
    norm = None if prev_norm is None else (prev_norm.weight, prev_norm.bias, prev_norm.eps)
    seed_a = _seed(p_attn, s_in.device)
    cv, attn, qkv, xn, mean0, rstd0 = KB.attn_block_fwd(s_in, norm, _w(f["w_qkv"]), f["b_qkv"], att.h, klen, causal, p_attn, seed_a)
    x = s_in if prev_norm is None else xn
    seed_r = _seed(p_res, s_in.device)
    s1 = _proj_res(cv.view(B * T, D), att.linear_out, x.view(B * T, D), p_res, seed_r)
    return s1, attn, (xn, mean0, rstd0, qkv, cv), (seed_a, seed_r)


def _dec_self_bwd(layer, prev_norm, s_in, saved, attn, dattn, seeds, p_res, p_attn, g_s1, ln1=None):
    """Backward of the self-attention block.  g_s1: gradient of s1 (ln1 None), or gradient of x1 = LN1(s1) with
    ln1 = (s1, mean1, rstd1, gamma1) -- then LN1' runs as the prologue of the block's kernel and ds1 is not needed outside.
    -> gradient of s_in."""
    xn, mean0, rstd0, qkv, cv = saved
    seed_a, seed_r = seeds
    B, T, D = s_in.shape
    M = B * T
    att = layer.self_attn
    f = att._fused
    dqkv = torch.empty_like(qkv)
    ds1, da = KB.attn_block_bwd(g_s1, ln1, None, p_res, 1.0, seed_r, _wt(att.linear_out.weight), qkv[..., :D], qkv[..., D:2 * D],
                                qkv[..., 2 * D:], attn, Fn._pad_like(dattn, attn), att.h, p_attn, seed_a, dqkv[..., :D], dqkv[..., D:2 * D],
                                dqkv[..., 2 * D:])
    if ln1 is None:
        ds1 = g_s1
    dqkv2 = dqkv.view(M, 3 * D)
    x = s_in if prev_norm is None else xn
    # x feeds the projection AND the residual: dx = dqkv Wqkv + ds1
    dx = _dgrad(dqkv2, f["w_qkv"], 3 * D, D, res=ds1.view(M, D))
    _wgrad(att.linear_out.weight, att.linear_out.bias, da.view(M, D), cv.view(M, D))
    _wgrad(f["w_qkv"], f["b_qkv"], dqkv2, x.view(M, D))
    if prev_norm is None:
        return dx.view(B, T, D)
    ds_in, _ = K.layernorm_bwd(dx, s_in.view(M, D), mean0, rstd0, prev_norm.weight)
    _norm_grads(prev_norm, dx, s_in.view(M, D), mean0, rstd0)
    return ds_in.view(B, T, D)


class _DecSelfBlock(Function):
    """The part of a post-LN decoder layer that does not see the memory (the decoder's head start, models/vtn.py):
    s1 = x + dropout(SelfAttention(x)), x = LN_prev(s_in) | s_in."""

    @staticmethod
    def forward(ctx, s_in, layer, prev_norm, klen, causal, p_res, p_attn):
        s_in = Fn._c(s_in)
        B, T, D = s_in.shape
        s1, attn, saved, seeds = _dec_self_fwd(s_in, prev_norm, layer, klen, causal, p_res, p_attn)
        ctx.mods = (layer, prev_norm)
        ctx.meta = (seeds, p_res, p_attn)
        ctx.n_saved = len(saved)
        ctx.save_for_backward(s_in, attn, *saved)
        ctx.set_materialize_grads(False)
        return s1.view(B, T, D), Fn._user_attn(attn, T)

    @staticmethod
    def backward(ctx, ds1, dattn):
        s_in, attn, *saved = ctx.saved_tensors
        layer, prev_norm = ctx.mods
        seeds, p_res, p_attn = ctx.meta
        if ds1 is None:
            ds1 = torch.zeros_like(s_in)
        ds_in = _dec_self_bwd(layer, prev_norm, s_in, tuple(saved), attn, dattn, seeds, p_res, p_attn, Fn._c(ds1))
        return ds_in, None, None, None, None, None, None


class _DecLayerPostLN(Function):
    """s3 of a post-LN DecoderLayer from the pre-norm stream of the previous layer (or, with `from_s1`, from the output of
    _DecSelfBlock): self-attention block, source-attention block over the packed K/V projection `kv` of the memory,
    feed-forward block; the layer's last LayerNorm (norm3) is left to the consumer."""

    @staticmethod
    def forward(ctx, s_in, kv, layer, prev_norm, from_s1, tgt_klen, mem_klen, causal, p_res, p_sa, p_ca, p_ffn):
        s_in = Fn._c(s_in)
        B, T, D = s_in.shape
        dev = s_in.device
        M = B * T
        ca, ff = layer.src_attn, layer.feed_forward
        fc = ca._fused
        if from_s1:
            s1, attn_s, saved_s, seeds_s = s_in.view(M, D), None, (), None
        else:
            s1, attn_s, saved_s, seeds_s = _dec_self_fwd(s_in, prev_norm, layer, tgt_klen, causal, p_res, p_sa)
        n1, n2 = layer.norm1, layer.norm2
        if not (kv.stride(-1) == 1 and KB._view_ok(kv)):
            kv = Fn._c(kv)
        seed_a = _seed(p_ca, dev)
        cv, attn_c, q2, x1, mean1, rstd1 = KB.attn_block_fwd(s1.view(B, T, D), (n1.weight, n1.bias, n1.eps), _w(fc["w_q"]), fc["b_q"], ca.h,
                                                             mem_klen, False, p_ca, seed_a, kv=(kv[..., :D], kv[..., D:]))
        seed_r2 = _seed(p_res, dev)
        s2 = _proj_res(cv.view(M, D), ca.linear_out, x1.view(M, D), p_res, seed_r2)
        Hd = ff.w_1.weight.shape[0]
        hmid = torch.empty((M, Hd), dtype=s_in.dtype, device=dev)
        seed_h = _seed(p_ffn, dev)
        x2, mean2, rstd2 = KB.gemm_rowpro(s2, _w(ff.w_1.weight), Hd, hmid, mode=1, norm=(n2.weight, n2.bias, n2.eps), bias=ff.w_1.bias,
                                          act="relu", drop_p=p_ffn, seed=seed_h)
        seed_r3 = _seed(p_res, dev)
        s3 = torch.empty((M, D), dtype=s_in.dtype, device=dev)
        K.gemm(K.operand(hmid, Hd), K.operand(_w(ff.w_2.weight), Hd), M, D, Hd, s3, in_dtype=s_in.dtype, bias=ff.w_2.bias, res=x2,
               drop_p=p_res, seed=seed_r3)
        ctx.mods = (layer, prev_norm)
        ctx.meta = (from_s1, p_res, p_sa, p_ca, p_ffn, seeds_s, seed_a, seed_r2, seed_h, seed_r3, (B, T, D))
        ctx.n_self = len(saved_s)
        ctx.save_for_backward(s_in, kv, attn_c, q2, cv, s1 if not from_s1 else None, x1, mean1, rstd1, s2, x2, mean2, rstd2, hmid,
                              attn_s, *saved_s)
        ctx.set_materialize_grads(False)
        return s3.view(B, T, D), Fn._user_attn(attn_c, kv.shape[1]), (Fn._user_attn(attn_s, T) if attn_s is not None else None)

    @staticmethod
    def backward(ctx, ds3, dattn_c, dattn_s):
        s_in, kv, attn_c, q2, cv, s1, x1, mean1, rstd1, s2, x2, mean2, rstd2, hmid, attn_s, *saved_s = ctx.saved_tensors
        layer, prev_norm = ctx.mods
        from_s1, p_res, p_sa, p_ca, p_ffn, seeds_s, seed_a, seed_r2, seed_h, seed_r3, (B, T, D) = ctx.meta
        ca, ff = layer.src_attn, layer.feed_forward
        fc = ca._fused
        n1, n2 = layer.norm1, layer.norm2
        M, Hd = B * T, hmid.shape[1]
        dev = s_in.device
        if from_s1:
            s1 = s_in.view(M, D)
        if ds3 is None:
            ds3 = torch.zeros((B, T, D), dtype=s_in.dtype, device=dev)
        g = Fn._c(ds3).view(M, D)
        # feed-forward block (x2 feeds it and the residual of s3): dx2 = du W1 + g
        du = torch.empty((M, Hd), dtype=s_in.dtype, device=dev)
        daf, _, _ = KB.gemm_rowpro(g, _wt(ff.w_2.weight), Hd, du, mode=2, p_a=p_res, seed_a=seed_r3, write_rows=True, emask=hmid,
                                   drop_p=p_ffn, seed=seed_h)
        dx2 = _dgrad(du, ff.w_1.weight, Hd, D, res=g)
        # source-attention block: ds2 = LN2'(dx2) ; da2 = ds2 * mask ; attention' -> dq, dk, dv
        dq = torch.empty_like(q2)
        dkv = torch.empty(kv.shape, dtype=kv.dtype, device=dev)
        ds2, da2 = KB.attn_block_bwd(dx2.view(B, T, D), (s2, mean2, rstd2, n2.weight), None, p_res, 1.0, seed_r2, _wt(ca.linear_out.weight),
                                     q2, kv[..., :D], kv[..., D:], attn_c, Fn._pad_like(dattn_c, attn_c), ca.h, p_ca, seed_a,
                                     dq, dkv[..., :D], dkv[..., D:])
        dq2 = dq.view(M, D)
        dx1 = _dgrad(dq2, fc["w_q"], D, D, res=ds2.view(M, D))          # x1 feeds the query projection and the residual of s2
        _wgrad(ff.w_2.weight, ff.w_2.bias, daf, hmid)
        _wgrad(ff.w_1.weight, ff.w_1.bias, du, x2)
        _norm_grads(n2, dx2, s2, mean2, rstd2)
        _wgrad(ca.linear_out.weight, ca.linear_out.bias, da2.view(M, D), cv.view(M, D))
        _wgrad(fc["w_q"], fc["b_q"], dq2, x1.view(M, D))
        _norm_grads(n1, dx1, s1, mean1, rstd1)
        if from_s1:
            ds1, _ = K.layernorm_bwd(dx1, s1, mean1, rstd1, n1.weight)
            return ds1.view(B, T, D), dkv, None, None, None, None, None, None, None, None, None, None
        # self-attention block, LN1' as the prologue of its kernel
        ds_in = _dec_self_bwd(layer, prev_norm, s_in, tuple(saved_s), attn_s, dattn_s, seeds_s, p_res, p_sa, dx1.view(B, T, D),
                              ln1=(s1, mean1, rstd1, n1.weight))
        return ds_in, dkv, None, None, None, None, None, None, None, None, None, None


def dec_self_block(layer, s_in, prev_norm, tgt_lens, causal):
    tr = layer.training
    s1, attn = _DecSelfBlock.apply(s_in, layer, prev_norm, None if tgt_lens is None else tgt_lens.dev, causal,
                                   layer.dropout_rate if tr else 0.0, layer.self_attn.dropout_rate if tr else 0.0)
    layer.self_attn.attn = attn
    return s1


def dec_layer(layer, s_in, prev_norm, tgt_lens, kv, mem_lens, causal, from_s1=False):
    """Pre-norm output stream s3 of a post-LN DecoderLayer (apply layer.norm3 -- or hand it to the next layer as prev_norm)."""
    tr = layer.training
    s3, attn_c, attn_s = _DecLayerPostLN.apply(s_in, kv, layer, prev_norm, from_s1, None if tgt_lens is None else tgt_lens.dev,
                                               None if mem_lens is None else mem_lens.dev, causal, layer.dropout_rate if tr else 0.0,
                                               layer.self_attn.dropout_rate if tr else 0.0, layer.src_attn.dropout_rate if tr else 0.0,
                                               layer.feed_forward.dropout_rate if tr else 0.0)
    layer.src_attn.attn = attn_c
    if attn_s is not None:
        layer.self_attn.attn = attn_s
    return s3
