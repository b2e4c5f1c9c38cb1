"""torch.ops.s2svc.* : the dispatcher-visible operator surface.

`load()` loads csrc/libs2svc_torch_ops.so (TORCH_LIBRARY(s2svc, ...) in csrc/torch_ops.cpp: schema strings + HIP implementations that
call the C ABI of include/s2svc_hip.h on torch's current stream) and attaches the autograd formulas of the differentiable ops with
`torch.library.register_autograd` -- each formula is one call of the matching `*_bwd` op, no Python arithmetic.  The package's own
modules do not go through these ops (their launchers bind the same C ABI through ctypes and batch / fuse across ops); this is the
boundary for callers that want plain torch operators: C++ / TorchScript / `torch.compile` users of the reference's call sites.
There is no CPU implementation: the ops raise NotImplementedError for CPU tensors.

    from seq2seq_vc_amd.ops import torch_library
    ops = torch_library.load()                     # == torch.ops.s2svc
    ds, path, bin_mean = ops.mas_forward(log_p_attn, text_lens, feat_lens)
"""
import torch

from .. import _lib

_LOADED = False

# name -> schema, as registered by csrc/torch_ops.cpp (tests compare this table with the dispatcher's)
SCHEMAS = {
    "abi_version": "s2svc::abi_version() -> int",
    "mas_forward": "s2svc::mas_forward(Tensor log_p_attn, Tensor text_lens, Tensor feat_lens) -> (Tensor ds, Tensor path, Tensor bin_mean)",
    "pairwise_l2_logsoftmax": "s2svc::pairwise_l2_logsoftmax(Tensor feats, Tensor text, Tensor text_lens) -> (Tensor log_p_attn, Tensor dist)",
    "pairwise_l2_logsoftmax_bwd": "s2svc::pairwise_l2_logsoftmax_bwd(Tensor log_p_attn, Tensor dist, Tensor dlogp, Tensor text_lens, ScalarType out_dtype) -> (Tensor G, Tensor rowsum)",
    "gaussian_upsample_probs": "s2svc::gaussian_upsample_probs(Tensor ds, Tensor text_lens, Tensor feat_lens, int T_feats, float delta, ScalarType out_dtype) -> Tensor",
    "betabinom_prior": "s2svc::betabinom_prior(Tensor text_lens, Tensor feat_lens, int T_feats, int T_text) -> Tensor",
    "ctc_forward_sum": "s2svc::ctc_forward_sum(Tensor log_p_attn, Tensor? prior, Tensor text_lens, Tensor feat_lens, float blank_logprob) -> (Tensor loss_per_utt, Tensor grad)",
    "masked_l1_bce": "s2svc::masked_l1_bce(Tensor? after, Tensor before, Tensor? logits, Tensor ys, Tensor? labels, Tensor olens, float pos_weight) -> Tensor",
    "masked_l1_bce_bwd": "s2svc::masked_l1_bce_bwd(Tensor? after, Tensor before, Tensor? logits, Tensor ys, Tensor? labels, Tensor olens, float pos_weight, Tensor stats, Tensor g_l1, Tensor g_bce) -> (Tensor d_after, Tensor d_before, Tensor d_logits)",
    "guided_attn_loss": "s2svc::guided_attn_loss(Tensor att, Tensor ilens, Tensor olens, float sigma, float alpha) -> Tensor",
    "guided_attn_loss_bwd": "s2svc::guided_attn_loss_bwd(Tensor att_like, Tensor ilens, Tensor olens, float sigma, float alpha, Tensor stats, Tensor gout) -> Tensor",
    "attn_fwd": "s2svc::attn_fwd(Tensor q, Tensor k, Tensor v, Tensor? klen, bool causal, int heads, float scale, float drop_p, Tensor? seed_base, int seed_off) -> (Tensor ctx, Tensor attn)",
    "attn_bwd": "s2svc::attn_bwd(Tensor q, Tensor k, Tensor v, Tensor dctx, Tensor attn, Tensor? dattn, int heads, float scale, float drop_p, Tensor? seed_base, int seed_off) -> (Tensor dq, Tensor dk, Tensor dv)",
    "ln_residual_dropout": "s2svc::ln_residual_dropout(Tensor x, Tensor? res, Tensor gamma, Tensor beta, float eps, float drop_p, float hscale, Tensor? seed_base, int seed_off) -> (Tensor y, Tensor s, Tensor mean, Tensor rstd)",
    "ln_residual_dropout_bwd": "s2svc::ln_residual_dropout_bwd(Tensor dy, Tensor s, Tensor mean, Tensor rstd, Tensor gamma, float drop_p, float hscale, Tensor? seed_base, int seed_off, bool has_res) -> (Tensor ds, Tensor dh)",
    "gemm_bias_act": "s2svc::gemm_bias_act(Tensor x, Tensor w, Tensor? bias, str act) -> Tensor",
    "batchnorm_stats": "s2svc::batchnorm_stats(Tensor x, float eps, float momentum, Tensor(a!)? run_mean, Tensor(b!)? run_var, Tensor(c!)? num_batches) -> (Tensor mean, Tensor rstd)",
}


def _register_autograd():
    ops = torch.ops.s2svc
    reg = torch.library.register_autograd

    # ---- ctc_forward_sum: the pass leaves `grad` = d (mean_b loss_per_utt[b]) / d log_p_attn (what forward_sum_loss.py:58-76 returns
    #      is that mean); only `loss_per_utt` is differentiable: d loss_per_utt[b] / d log_p_attn[b] = B * grad[b] ------------------
    def fs_setup(ctx, inputs, output):
        ctx.save_for_backward(output[1])

    def fs_bwd(ctx, g_loss, _g_grad):
        (grad,) = ctx.saved_tensors
        return grad * (g_loss.view(-1, 1, 1) * float(grad.shape[0])), None, None, None, None
    reg("s2svc::ctc_forward_sum", fs_bwd, setup_context=fs_setup)

    # ---- masked_l1_bce -> stats (l1, bce, count): gradients of the first two -----------------------------------------------------
    def sl_setup(ctx, inputs, output):
        after, before, logits, ys, labels, olens, pos_weight = inputs
        ctx.pos_weight = pos_weight
        ctx.has = (after is not None, logits is not None)
        ctx.save_for_backward(after, before, logits, ys, labels, olens, output)

    def sl_bwd(ctx, g):
        after, before, logits, ys, labels, olens, stats = ctx.saved_tensors
        g = g.float().contiguous()
        da, db, dl = ops.masked_l1_bce_bwd(after, before, logits, ys, labels, olens, ctx.pos_weight, stats, g[0:1].clone(), g[1:2].clone())
        return (da if ctx.has[0] else None), db, (dl if ctx.has[1] else None), None, None, None, None
    reg("s2svc::masked_l1_bce", sl_bwd, setup_context=sl_setup)

    # ---- guided_attn_loss -> stats (loss, count) ------------------------------------------------------------------------------
    def ga_setup(ctx, inputs, output):
        att, ilens, olens, sigma, alpha = inputs
        ctx.sa = (sigma, alpha)
        ctx.save_for_backward(att, ilens, olens, output)

    def ga_bwd(ctx, g):
        att, ilens, olens, stats = ctx.saved_tensors
        return ops.guided_attn_loss_bwd(att, ilens, olens, ctx.sa[0], ctx.sa[1], stats, g.float()[0:1].clone()), None, None, None, None
    reg("s2svc::guided_attn_loss", ga_bwd, setup_context=ga_setup)

    # ---- attn_fwd: gradients of q, k, v from (d ctx, d attn) -------------------------------------------------------------------
    def at_setup(ctx, inputs, output):
        q, k, v, klen, causal, heads, scale, drop_p, seed_base, seed_off = inputs
        ctx.meta = (heads, scale, drop_p, seed_off)
        ctx.save_for_backward(q, k, v, output[1], seed_base)

    def at_bwd(ctx, dctx, dattn):
        q, k, v, attn, seed_base = ctx.saved_tensors
        heads, scale, drop_p, seed_off = ctx.meta
        if dctx is None:
            dctx = torch.zeros_like(q)
        dq, dk, dv = ops.attn_bwd(q, k, v, dctx.contiguous(), attn, None if dattn is None else dattn.contiguous(), heads, scale, drop_p,
                                  seed_base, seed_off)
        return dq, dk, dv, None, None, None, None, None, None, None
    reg("s2svc::attn_fwd", at_bwd, setup_context=at_setup)

    # ---- ln_residual_dropout: gradients of x and res (y only; gamma / beta gradients are column reductions the package batches) ----
    def ln_setup(ctx, inputs, output):
        x, res, gamma, beta, eps, drop_p, hscale, seed_base, seed_off = inputs
        ctx.meta = (drop_p, hscale, seed_off, res is not None)
        ctx.save_for_backward(output[1] if res is not None else x, output[2], output[3], gamma, seed_base)      # s (= x without a residual)

    def ln_bwd(ctx, dy, _ds, _dm, _dr):
        s, mean, rstd, gamma, seed_base = ctx.saved_tensors
        drop_p, hscale, seed_off, has_res = ctx.meta
        ds, dh = ops.ln_residual_dropout_bwd(dy.contiguous(), s, mean, rstd, gamma, drop_p, hscale, seed_base, seed_off, has_res)
        return dh, (ds if has_res else None), None, None, None, None, None, None, None
    reg("s2svc::ln_residual_dropout", ln_bwd, setup_context=ln_setup)


def load():
    """Load the registration library (built by `_lib.build_torch_ops()`, i.e. `__graft_entry__.build()`) once; -> torch.ops.s2svc."""
    global _LOADED
    if not _LOADED:
        import os
        if not os.path.exists(_lib.TORCH_OPS_LIB_PATH):
            raise RuntimeError(f"{_lib.TORCH_OPS_LIB_PATH} is missing: run __graft_entry__.build() (or seq2seq_vc_amd._lib.build_torch_ops())")
        _lib.lib()                                     # libs2svc_hip.so first (the registration library links against it)
        torch.ops.load_library(_lib.TORCH_OPS_LIB_PATH)
        _register_autograd()
        _LOADED = True
    return torch.ops.s2svc
