// torch.ops.s2svc.* -- the operator surface SURVEY.md section 8(b) asks for: the hot-path kernels of libs2svc_hip.so registered with the
// torch dispatcher (TORCH_LIBRARY: the schema strings below are the signature contract), implemented for the HIP device only by
// calling the extern "C" entry points of include/s2svc_hip.h on torch's current stream.  There is NO CPU implementation: a call on CPU
// tensors fails in the dispatcher ("no kernel for the CPU backend"), as the tier rules ask of a product path without a fallback.
// Host-only C++ (no kernels): compiled with g++ against the torch headers by _lib.build_torch_ops(); the Python package does not
// need it (its launchers bind the same C ABI through ctypes) -- this library is the boundary a C++ / TorchScript / torch.compile
// caller of the reference would bind, see INTEGRATION.md "level 3".  Autograd formulas for the differentiable ops are attached in
// seq2seq_vc_amd/ops/torch_library.py (torch.library.register_autograd over the *_bwd ops defined here).
//
// Reference call sites each op replaces are listed at the declaration of its entry point in include/s2svc_hip.h.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <string>

#include "../../include/s2svc_hip.h"

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc >= 0, "s2svc::", what, " failed: ", s2svc_last_error());
}
int dt_code(const at::Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16, "s2svc ops take float32 or bfloat16 tensors");
  return t.scalar_type() == at::kFloat ? 0 : 1;
}
int dt_code(at::ScalarType st) {
  TORCH_CHECK(st == at::kFloat || st == at::kBFloat16, "s2svc ops produce float32 or bfloat16 tensors");
  return st == at::kFloat ? 0 : 1;
}
const at::Tensor& need(const at::Tensor& t, at::ScalarType st, const char* name) {
  TORCH_CHECK(t.is_cuda(), "s2svc: `", name, "` must live on the GPU (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == st, "s2svc: `", name, "` must be ", st);
  TORCH_CHECK(t.is_contiguous(), "s2svc: `", name, "` must be contiguous");
  return t;
}
const at::Tensor& need_act(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "s2svc: `", name, "` must live on the GPU (there is no CPU path)");
  TORCH_CHECK(t.is_contiguous(), "s2svc: `", name, "` must be contiguous");
  dt_code(t);
  return t;
}
at::Tensor lens_i32(const at::Tensor& l, const at::Device& dev, const char* name) {       // (B,) int32 / int64, any device -> int32 on dev
  TORCH_CHECK(l.dim() == 1 && (l.scalar_type() == at::kInt || l.scalar_type() == at::kLong), "s2svc: `", name, "` must be a 1-D int32 / int64 tensor");
  return l.to(dev, at::kInt).contiguous();
}
const void* optp(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

// ---- alignment search ------------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor, at::Tensor> mas_forward(const at::Tensor& log_p_attn, const at::Tensor& text_lens,
                                                           const at::Tensor& feat_lens) {
  need(log_p_attn, at::kFloat, "log_p_attn");
  TORCH_CHECK(log_p_attn.dim() == 3, "mas_forward: log_p_attn is (B, T_feats, T_text)");
  const int B = log_p_attn.size(0), Tf = log_p_attn.size(1), Tx = log_p_attn.size(2);
  const auto dev = log_p_attn.device();
  const at::Tensor tl = lens_i32(text_lens, dev, "text_lens"), fl = lens_i32(feat_lens, dev, "feat_lens");
  at::Tensor ds = at::empty({B, Tx}, log_p_attn.options());
  at::Tensor path = at::empty({B, Tf}, log_p_attn.options().dtype(at::kInt));
  at::Tensor binmean = at::empty({B}, log_p_attn.options());
  at::Tensor ws = at::empty({s2svc_mas_ws_bytes(B, Tf, Tx) / 8 + 1}, log_p_attn.options().dtype(at::kLong));
  check_rc(s2svc_mas(B, Tf, Tx, log_p_attn.data_ptr<float>(), tl.data_ptr<int32_t>(), fl.data_ptr<int32_t>(), path.data_ptr<int32_t>(),
                     ds.data_ptr<float>(), binmean.data_ptr<float>(), ws.data_ptr(), cur_stream()), "mas_forward");
  return {ds, path, binmean};
}

std::tuple<at::Tensor, at::Tensor> pairwise_l2_logsoftmax(const at::Tensor& feats, const at::Tensor& text, const at::Tensor& text_lens) {
  need_act(feats, "feats");
  need_act(text, "text");
  TORCH_CHECK(feats.dim() == 3 && text.dim() == 3 && feats.size(0) == text.size(0) && feats.size(2) == text.size(2) &&
              feats.scalar_type() == text.scalar_type(), "pairwise_l2_logsoftmax: feats (B, T_f, A), text (B, T_x, A) of one dtype");
  const int B = feats.size(0), Tf = feats.size(1), Tx = text.size(1), A = feats.size(2);
  const at::Tensor tl = lens_i32(text_lens, feats.device(), "text_lens");
  at::Tensor logp = at::empty({B, Tf, Tx}, feats.options().dtype(at::kFloat));
  at::Tensor dist = at::empty({B, Tf, Tx}, feats.options().dtype(at::kFloat));
  check_rc(s2svc_pairwise_l2_logsoftmax(dt_code(feats), B, Tf, Tx, A, feats.data_ptr(), text.data_ptr(), tl.data_ptr<int32_t>(),
                                        logp.data_ptr<float>(), dist.data_ptr<float>(), cur_stream()), "pairwise_l2_logsoftmax");
  return {logp, dist};
}

// G[b, t, j] = (softmax-backward of dlogp)[b, t, j] / dist[b, t, j] and its row sums: d feats = rowsum * feats - G . text,
// d text = G^T . feats - colsum(G) * text (ops/functional_aas.py runs those products on the GEMM kernels)
std::tuple<at::Tensor, at::Tensor> pairwise_l2_logsoftmax_bwd(const at::Tensor& logp, const at::Tensor& dist, const at::Tensor& dlogp,
                                                              const at::Tensor& text_lens, at::ScalarType out_dtype) {
  need(logp, at::kFloat, "logp");
  need(dist, at::kFloat, "dist");
  need(dlogp, at::kFloat, "dlogp");
  const int B = logp.size(0), Tf = logp.size(1), Tx = logp.size(2);
  const at::Tensor tl = lens_i32(text_lens, logp.device(), "text_lens");
  at::Tensor G = at::empty({B, Tf, Tx}, logp.options().dtype(out_dtype));
  at::Tensor rowsum = at::empty({B, Tf}, logp.options());
  check_rc(s2svc_pairwise_l2_bwd_g(dt_code(out_dtype), B, Tf, Tx, logp.data_ptr<float>(), dist.data_ptr<float>(), dlogp.data_ptr<float>(),
                                   tl.data_ptr<int32_t>(), G.data_ptr(), rowsum.data_ptr<float>(), cur_stream()), "pairwise_l2_logsoftmax_bwd");
  return {G, rowsum};
}

at::Tensor gaussian_upsample_probs(const at::Tensor& ds, const at::Tensor& text_lens, const at::Tensor& feat_lens, int64_t Tf, double delta,
                                   at::ScalarType out_dtype) {
  need(ds, at::kFloat, "ds");
  const int B = ds.size(0), Tx = ds.size(1);
  const at::Tensor tl = lens_i32(text_lens, ds.device(), "text_lens"), fl = lens_i32(feat_lens, ds.device(), "feat_lens");
  at::Tensor P = at::empty({B, Tf, Tx}, ds.options().dtype(out_dtype));
  check_rc(s2svc_gauss_upsample_probs(dt_code(out_dtype), B, (int)Tf, Tx, ds.data_ptr<float>(), tl.data_ptr<int32_t>(), fl.data_ptr<int32_t>(),
                                      (float)delta, P.data_ptr(), cur_stream()), "gaussian_upsample_probs");
  return P;
}

at::Tensor betabinom_prior(const at::Tensor& text_lens, const at::Tensor& feat_lens, int64_t Tf, int64_t Tx) {
  TORCH_CHECK(text_lens.is_cuda() && feat_lens.is_cuda(), "betabinom_prior: the length vectors name the device; they must live on the GPU");
  const at::Tensor tl = lens_i32(text_lens, text_lens.device(), "text_lens"), fl = lens_i32(feat_lens, text_lens.device(), "feat_lens");
  const int B = tl.size(0);
  at::Tensor prior = at::empty({B, Tf, Tx}, tl.options().dtype(at::kFloat));
  check_rc(s2svc_betabinom_prior(B, (int)Tf, (int)Tx, tl.data_ptr<int32_t>(), fl.data_ptr<int32_t>(), prior.data_ptr<float>(), cur_stream()),
           "betabinom_prior");
  return prior;
}

// per-utterance forward-sum (CTC with targets 1..N) loss and its gradient wrt log_p_attn in one pass
std::tuple<at::Tensor, at::Tensor> ctc_forward_sum(const at::Tensor& log_p_attn, const c10::optional<at::Tensor>& prior,
                                                   const at::Tensor& text_lens, const at::Tensor& feat_lens, double blank_logprob) {
  need(log_p_attn, at::kFloat, "log_p_attn");
  if (prior.has_value()) need(*prior, at::kFloat, "prior");
  const int B = log_p_attn.size(0), Tf = log_p_attn.size(1), Tx = log_p_attn.size(2);
  const auto dev = log_p_attn.device();
  const at::Tensor tl = lens_i32(text_lens, dev, "text_lens"), fl = lens_i32(feat_lens, dev, "feat_lens");
  at::Tensor ws = at::empty({s2svc_forward_sum_ws_bytes(B, Tf, Tx) / 4 + 1}, log_p_attn.options());
  at::Tensor loss_b = at::empty({B}, log_p_attn.options());
  at::Tensor grad = at::empty({B, Tf, Tx}, log_p_attn.options());
  check_rc(s2svc_forward_sum(B, Tf, Tx, log_p_attn.data_ptr<float>(), (const float*)optp(prior), tl.data_ptr<int32_t>(), fl.data_ptr<int32_t>(),
                             (float)blank_logprob, ws.data_ptr(), loss_b.data_ptr<float>(), grad.data_ptr<float>(), cur_stream()),
           "ctc_forward_sum");
  return {loss_b, grad};
}

// ---- losses ----------------------------------------------------------------------------------------------------------------
// -> stats (3) fp32: [mean L1(after) + mean L1(before), BCE-with-logits (pos_weight), number of valid frames]
at::Tensor masked_l1_bce(const c10::optional<at::Tensor>& after, const at::Tensor& before, const c10::optional<at::Tensor>& logits,
                         const at::Tensor& ys, const c10::optional<at::Tensor>& labels, const at::Tensor& olens, double pos_weight) {
  need_act(before, "before");
  need(ys, at::kFloat, "ys");
  if (after.has_value()) need_act(*after, "after");
  if (logits.has_value()) need_act(*logits, "logits");
  if (labels.has_value()) need(*labels, at::kFloat, "labels");
  TORCH_CHECK(before.dim() == 3, "masked_l1_bce: before is (B, T, odim)");
  const int B = before.size(0), Tm = before.size(1), D = before.size(2);
  const at::Tensor ol = lens_i32(olens, before.device(), "olens");
  at::Tensor partial = at::empty({3 * 1024}, before.options().dtype(at::kFloat));
  at::Tensor out = at::empty({3}, before.options().dtype(at::kFloat));
  check_rc(s2svc_seq_loss_fwd(dt_code(before), B, Tm, D, optp(after), before.data_ptr(), optp(logits), ys.data_ptr<float>(),
                              (const float*)optp(labels), ol.data_ptr<int32_t>(), (float)pos_weight, partial.data_ptr<float>(),
                              out.data_ptr<float>(), cur_stream()), "masked_l1_bce");
  return out;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> masked_l1_bce_bwd(const c10::optional<at::Tensor>& after, const at::Tensor& before,
                                                                 const c10::optional<at::Tensor>& logits, const at::Tensor& ys,
                                                                 const c10::optional<at::Tensor>& labels, const at::Tensor& olens,
                                                                 double pos_weight, const at::Tensor& stats, const at::Tensor& g_l1,
                                                                 const at::Tensor& g_bce) {
  need_act(before, "before");
  need(stats, at::kFloat, "stats");
  need(g_l1, at::kFloat, "g_l1");
  need(g_bce, at::kFloat, "g_bce");
  const int B = before.size(0), Tm = before.size(1), D = before.size(2);
  const at::Tensor ol = lens_i32(olens, before.device(), "olens");
  at::Tensor d_after = after.has_value() ? at::empty_like(*after) : at::Tensor();
  at::Tensor d_before = at::empty_like(before);
  at::Tensor d_logits = logits.has_value() ? at::empty_like(*logits) : at::Tensor();
  check_rc(s2svc_seq_loss_bwd(dt_code(before), B, Tm, D, optp(after), before.data_ptr(), optp(logits), ys.data_ptr<float>(),
                              (const float*)optp(labels), ol.data_ptr<int32_t>(), (float)pos_weight, stats.data_ptr<float>(),
                              g_l1.data_ptr<float>(), g_bce.data_ptr<float>(), d_after.defined() ? d_after.data_ptr() : nullptr,
                              d_before.data_ptr(), d_logits.defined() ? d_logits.data_ptr() : nullptr, cur_stream()), "masked_l1_bce_bwd");
  return {d_after.defined() ? d_after : at::zeros({0}, before.options()), d_before,
          d_logits.defined() ? d_logits : at::zeros({0}, before.options())};
}

// -> stats (2) fp32: [alpha * mean over valid (t, n) of W * att, number of valid elements]
at::Tensor guided_attn_loss(const at::Tensor& att, const at::Tensor& ilens, const at::Tensor& olens, double sigma, double alpha) {
  need_act(att, "att");
  TORCH_CHECK(att.dim() == 4, "guided_attn_loss: att is (B, H, T_out, T_in)");
  const int B = att.size(0), H = att.size(1), To = att.size(2), Ti = att.size(3);
  const at::Tensor il = lens_i32(ilens, att.device(), "ilens"), ol = lens_i32(olens, att.device(), "olens");
  at::Tensor partial = at::empty({1024}, att.options().dtype(at::kFloat));
  at::Tensor out = at::empty({2}, att.options().dtype(at::kFloat));
  check_rc(s2svc_guided_attn_loss_fwd(dt_code(att), B, H, To, Ti, att.data_ptr(), il.data_ptr<int32_t>(), ol.data_ptr<int32_t>(), (float)sigma,
                                      (float)alpha, partial.data_ptr<float>(), out.data_ptr<float>(), cur_stream()), "guided_attn_loss");
  return out;
}

at::Tensor guided_attn_loss_bwd(const at::Tensor& att_like, const at::Tensor& ilens, const at::Tensor& olens, double sigma, double alpha,
                                const at::Tensor& stats, const at::Tensor& gout) {
  need_act(att_like, "att_like");
  need(stats, at::kFloat, "stats");
  need(gout, at::kFloat, "gout");
  const int B = att_like.size(0), H = att_like.size(1), To = att_like.size(2), Ti = att_like.size(3);
  const at::Tensor il = lens_i32(ilens, att_like.device(), "ilens"), ol = lens_i32(olens, att_like.device(), "olens");
  at::Tensor datt = at::empty_like(att_like);
  check_rc(s2svc_guided_attn_loss_bwd(dt_code(att_like), B, H, To, Ti, il.data_ptr<int32_t>(), ol.data_ptr<int32_t>(), (float)sigma, (float)alpha,
                                      stats.data_ptr<float>(), gout.data_ptr<float>(), datt.data_ptr(), cur_stream()), "guided_attn_loss_bwd");
  return datt;
}

// ---- attention (short sequences, bf16: one launch forward, one backward) -------------------------------------------------------
bool view3_ok(const at::Tensor& t) { return t.dim() == 3 && t.stride(2) == 1 && t.stride(1) % 8 == 0 && t.stride(0) % 8 == 0 && ((uintptr_t)t.data_ptr()) % 16 == 0; }

std::tuple<at::Tensor, at::Tensor> attn_fwd(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, const c10::optional<at::Tensor>& klen,
                                            bool causal, int64_t heads, double scale, double drop_p, const c10::optional<at::Tensor>& seed_base,
                                            int64_t seed_off) {
  TORCH_CHECK(q.is_cuda() && q.scalar_type() == at::kBFloat16 && k.scalar_type() == at::kBFloat16 && v.scalar_type() == at::kBFloat16,
              "attn_fwd: bf16 GPU tensors (the fp32 parity mode runs the separate GEMM + softmax kernels)");
  TORCH_CHECK(view3_ok(q) && view3_ok(k) && view3_ok(v), "attn_fwd: (B, T, D) views with contiguous features, strides multiples of 8, 16-byte aligned");
  const int B = q.size(0), T1 = q.size(1), T2 = k.size(1), D = q.size(2), H = (int)heads, dk = D / H;
  TORCH_CHECK(s2svc_attn_fused_supported(1, T1, T2, dk), "attn_fwd: T1, T2 <= 64 and d_k in {32, 64, 96, 128}");
  const int ld = (T2 + 7) / 8 * 8;
  at::Tensor out = at::empty({B, T1, D}, q.options());
  at::Tensor attn = at::empty({B, H, T1, ld}, q.options());
  at::Tensor kl;
  if (klen.has_value()) kl = lens_i32(*klen, q.device(), "klen");
  check_rc(s2svc_attn_fused_fwd(B, H, T1, T2, dk, q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0), v.data_ptr(),
                                v.stride(1), v.stride(0), kl.defined() ? kl.data_ptr<int32_t>() : nullptr, causal ? 1 : 0, (float)scale,
                                (float)drop_p, (const uint64_t*)optp(seed_base), (uint64_t)seed_off, attn.data_ptr(), ld, out.data_ptr(), D,
                                (int64_t)T1 * D, cur_stream()), "attn_fwd");
  return {out, attn};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> attn_bwd(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, const at::Tensor& dout,
                                                        const at::Tensor& attn, const c10::optional<at::Tensor>& dattn, int64_t heads,
                                                        double scale, double drop_p, const c10::optional<at::Tensor>& seed_base,
                                                        int64_t seed_off) {
  TORCH_CHECK(view3_ok(q) && view3_ok(k) && view3_ok(v) && view3_ok(dout), "attn_bwd: (B, T, D) views, strides multiples of 8, 16-byte aligned");
  need(attn, at::kBFloat16, "attn");
  const int B = q.size(0), T1 = q.size(1), T2 = k.size(1), D = q.size(2), H = (int)heads, dk = D / H, ld = attn.size(3);
  at::Tensor dq = at::empty({B, T1, D}, q.options()), dkk = at::empty({B, T2, D}, q.options()), dv = at::empty({B, T2, D}, q.options());
  check_rc(s2svc_attn_fused_bwd(B, H, T1, T2, dk, q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0), v.data_ptr(),
                                v.stride(1), v.stride(0), dout.data_ptr(), dout.stride(1), dout.stride(0), attn.data_ptr(), optp(dattn), ld,
                                (float)scale, (float)drop_p, (const uint64_t*)optp(seed_base), (uint64_t)seed_off, dq.data_ptr(), D,
                                (int64_t)T1 * D, dkk.data_ptr(), D, (int64_t)T2 * D, dv.data_ptr(), D, (int64_t)T2 * D, cur_stream()), "attn_bwd");
  return {dq, dkk, dv};
}

// ---- residual + dropout + LayerNorm ----------------------------------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> ln_residual_dropout(const at::Tensor& x, const c10::optional<at::Tensor>& res,
                                                                               const at::Tensor& gamma, const at::Tensor& beta, double eps,
                                                                               double drop_p, double hscale,
                                                                               const c10::optional<at::Tensor>& seed_base, int64_t seed_off) {
  need_act(x, "x");
  need(gamma, at::kFloat, "gamma");
  need(beta, at::kFloat, "beta");
  if (res.has_value()) { need_act(*res, "res"); TORCH_CHECK(res->scalar_type() == x.scalar_type() && res->sizes() == x.sizes(), "ln_residual_dropout: res like x"); }
  const int D = x.size(-1), rows = (int)(x.numel() / D);
  at::Tensor y = at::empty_like(x);
  at::Tensor s = res.has_value() ? at::empty_like(x) : at::empty({0}, x.options());       // no residual: the LayerNorm input IS x (no copy, no alias returned)
  at::Tensor mean = at::empty({rows}, x.options().dtype(at::kFloat)), rstd = at::empty({rows}, x.options().dtype(at::kFloat));
  check_rc(s2svc_layernorm_fwd(dt_code(x), rows, D, x.data_ptr(), optp(res), (float)drop_p, (float)hscale, (const uint64_t*)optp(seed_base),
                               (uint64_t)seed_off, gamma.data_ptr<float>(), beta.data_ptr<float>(), (float)eps, y.data_ptr(),
                               res.has_value() ? s.data_ptr() : nullptr, mean.data_ptr<float>(), rstd.data_ptr<float>(), cur_stream()),
           "ln_residual_dropout");
  return {y, s, mean, rstd};
}

// -> (ds = gradient wrt the LayerNorm input s (= gradient wrt res), dh = gradient wrt x (ds * dropout mask * hscale))
std::tuple<at::Tensor, at::Tensor> ln_residual_dropout_bwd(const at::Tensor& dy, const at::Tensor& s, const at::Tensor& mean, const at::Tensor& rstd,
                                                           const at::Tensor& gamma, double drop_p, double hscale,
                                                           const c10::optional<at::Tensor>& seed_base, int64_t seed_off, bool has_res) {
  need_act(dy, "dy");
  need_act(s, "s");
  need(mean, at::kFloat, "mean");
  need(rstd, at::kFloat, "rstd");
  need(gamma, at::kFloat, "gamma");
  const int D = s.size(-1), rows = (int)(s.numel() / D);
  at::Tensor ds = at::empty_like(s);
  at::Tensor dh = has_res ? at::empty_like(s) : ds;
  check_rc(s2svc_layernorm_bwd(dt_code(s), rows, D, dy.data_ptr(), s.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                               gamma.data_ptr<float>(), nullptr, (float)drop_p, (float)hscale, (const uint64_t*)optp(seed_base), (uint64_t)seed_off,
                               ds.data_ptr(), has_res ? dh.data_ptr() : nullptr, cur_stream()), "ln_residual_dropout_bwd");
  return {ds, dh};
}

// ---- Linear: y = act(x W^T + b) on the MFMA GEMM kernels ---------------------------------------------------------------------------
int act_code(const std::string& a) {
  if (a.empty() || a == "none") return 0;
  if (a == "relu") return 1;
  if (a == "tanh") return 2;
  if (a == "swish") return 3;
  if (a == "sigmoid") return 4;
  if (a == "gelu") return 5;
  TORCH_CHECK(false, "gemm_bias_act: unknown activation '", a, "'");
}

at::Tensor gemm_bias_act(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, std::string act) {
  need_act(x, "x");
  need_act(w, "w");
  TORCH_CHECK(w.dim() == 2 && x.size(-1) == w.size(1) && x.scalar_type() == w.scalar_type(), "gemm_bias_act: x (..., K), w (N, K) of one dtype");
  if (bias.has_value()) need(*bias, at::kFloat, "bias");
  const int64_t K = w.size(1), N = w.size(0), M = x.numel() / K;
  auto sizes = x.sizes().vec();
  sizes.back() = N;
  at::Tensor y = at::empty(sizes, x.options());
  s2svc_gemm_desc d;
  std::memset(&d, 0, sizeof(d));
  d.A.ptr = x.data_ptr(); d.A.ld = K; d.A.layout = S2SVC_LAYOUT_KC; d.A.mode = S2SVC_OP_DENSE;
  d.B.ptr = w.data_ptr(); d.B.ld = K; d.B.layout = S2SVC_LAYOUT_KC; d.B.mode = S2SVC_OP_DENSE;
  d.C = y.data_ptr(); d.ldc = N; d.c_dtype = dt_code(x); d.dtype = dt_code(x);
  d.bias = (const float*)optp(bias);
  d.M = (int)M; d.N = (int)N; d.K = (int)K; d.nb0 = d.nb1 = 1; d.alpha = 1.0f; d.act = act_code(act);
  check_rc(s2svc_gemm(&d, cur_stream()), "gemm_bias_act");
  return y;
}

// ---- BatchNorm1d training statistics over (rows, C) channel-last ---------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor> batchnorm_stats(const at::Tensor& x, double eps, double momentum, const c10::optional<at::Tensor>& run_mean,
                                                   const c10::optional<at::Tensor>& run_var, const c10::optional<at::Tensor>& num_batches) {
  need_act(x, "x");
  const int C = x.size(-1), rows = (int)(x.numel() / C);
  if (run_mean.has_value()) need(*run_mean, at::kFloat, "run_mean");
  if (run_var.has_value()) need(*run_var, at::kFloat, "run_var");
  if (num_batches.has_value()) need(*num_batches, at::kLong, "num_batches");
  const int chunks = 64;
  at::Tensor mean = at::empty({C}, x.options().dtype(at::kFloat)), rstd = at::empty({C}, x.options().dtype(at::kFloat));
  at::Tensor ws = at::empty({(int64_t)chunks * 2 * C}, x.options().dtype(at::kFloat));
  check_rc(s2svc_bn_stats(dt_code(x), rows, C, x.data_ptr(), (float)eps, (float)momentum, mean.data_ptr<float>(), rstd.data_ptr<float>(),
                          (float*)optp(run_mean), (float*)optp(run_var), (int64_t*)optp(num_batches), ws.data_ptr<float>(), chunks, 0, nullptr,
                          cur_stream()), "batchnorm_stats");
  return {mean, rstd};
}

int64_t abi_version() { return s2svc_abi_version(); }

}  // namespace

TORCH_LIBRARY(s2svc, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("mas_forward(Tensor log_p_attn, Tensor text_lens, Tensor feat_lens) -> (Tensor ds, Tensor path, Tensor bin_mean)");
  m.def("pairwise_l2_logsoftmax(Tensor feats, Tensor text, Tensor text_lens) -> (Tensor log_p_attn, Tensor dist)");
  m.def("pairwise_l2_logsoftmax_bwd(Tensor log_p_attn, Tensor dist, Tensor dlogp, Tensor text_lens, ScalarType out_dtype) -> (Tensor G, Tensor rowsum)");
  m.def("gaussian_upsample_probs(Tensor ds, Tensor text_lens, Tensor feat_lens, int T_feats, float delta, ScalarType out_dtype) -> Tensor");
  m.def("betabinom_prior(Tensor text_lens, Tensor feat_lens, int T_feats, int T_text) -> Tensor");
  m.def("ctc_forward_sum(Tensor log_p_attn, Tensor? prior, Tensor text_lens, Tensor feat_lens, float blank_logprob) -> (Tensor loss_per_utt, Tensor grad)");
  m.def("masked_l1_bce(Tensor? after, Tensor before, Tensor? logits, Tensor ys, Tensor? labels, Tensor olens, float pos_weight) -> Tensor");
  m.def("masked_l1_bce_bwd(Tensor? after, Tensor before, Tensor? logits, Tensor ys, Tensor? labels, Tensor olens, float pos_weight, Tensor stats, "
        "Tensor g_l1, Tensor g_bce) -> (Tensor d_after, Tensor d_before, Tensor d_logits)");
  m.def("guided_attn_loss(Tensor att, Tensor ilens, Tensor olens, float sigma, float alpha) -> Tensor");
  m.def("guided_attn_loss_bwd(Tensor att_like, Tensor ilens, Tensor olens, float sigma, float alpha, Tensor stats, Tensor gout) -> Tensor");
  m.def("attn_fwd(Tensor q, Tensor k, Tensor v, Tensor? klen, bool causal, int heads, float scale, float drop_p, Tensor? seed_base, int seed_off) -> "
        "(Tensor ctx, Tensor attn)");
  m.def("attn_bwd(Tensor q, Tensor k, Tensor v, Tensor dctx, Tensor attn, Tensor? dattn, int heads, float scale, float drop_p, Tensor? seed_base, "
        "int seed_off) -> (Tensor dq, Tensor dk, Tensor dv)");
  m.def("ln_residual_dropout(Tensor x, Tensor? res, Tensor gamma, Tensor beta, float eps, float drop_p, float hscale, Tensor? seed_base, int seed_off) "
        "-> (Tensor y, Tensor s, Tensor mean, Tensor rstd)");
  m.def("ln_residual_dropout_bwd(Tensor dy, Tensor s, Tensor mean, Tensor rstd, Tensor gamma, float drop_p, float hscale, Tensor? seed_base, "
        "int seed_off, bool has_res) -> (Tensor ds, Tensor dh)");
  m.def("gemm_bias_act(Tensor x, Tensor w, Tensor? bias, str act) -> Tensor");
  m.def("batchnorm_stats(Tensor x, float eps, float momentum, Tensor(a!)? run_mean, Tensor(b!)? run_var, Tensor(c!)? num_batches) -> (Tensor mean, Tensor rstd)");
}

TORCH_LIBRARY_IMPL(s2svc, CUDA, m) {       // (the dispatch key of HIP devices in a ROCm build of torch)
  m.impl("mas_forward", &mas_forward);
  m.impl("pairwise_l2_logsoftmax", &pairwise_l2_logsoftmax);
  m.impl("pairwise_l2_logsoftmax_bwd", &pairwise_l2_logsoftmax_bwd);
  m.impl("gaussian_upsample_probs", &gaussian_upsample_probs);
  m.impl("betabinom_prior", &betabinom_prior);
  m.impl("ctc_forward_sum", &ctc_forward_sum);
  m.impl("masked_l1_bce", &masked_l1_bce);
  m.impl("masked_l1_bce_bwd", &masked_l1_bce_bwd);
  m.impl("guided_attn_loss", &guided_attn_loss);
  m.impl("guided_attn_loss_bwd", &guided_attn_loss_bwd);
  m.impl("attn_fwd", &attn_fwd);
  m.impl("attn_bwd", &attn_bwd);
  m.impl("ln_residual_dropout", &ln_residual_dropout);
  m.impl("ln_residual_dropout_bwd", &ln_residual_dropout_bwd);
  m.impl("gemm_bias_act", &gemm_bias_act);
  m.impl("batchnorm_stats", &batchnorm_stats);
}
