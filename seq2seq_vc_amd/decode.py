"""Autoregressive decoding with a static K/V cache and one captured hipGraph per step.

Replaces the generation loop of the reference (models/vtn.py:334-394, models/transformer_tts.py:258-326) and
`Decoder.forward_one_step` (modules/transformer/decoder.py:239-273, decoder_layer.py:85-132):

* reference: every step re-embeds the WHOLE prefix (prenet + positional encoding), re-projects K/V of the prefix
  in every layer, re-projects the source K/V in every layer, grows `ys` with `torch.cat`, walks
  `named_modules()` for the attention weights and reads the stop probability on the host (one sync per step);
* here: source K/V are projected once, self-attention K/V are appended to a static cache, the step index, the
  per-utterance stop test and the dropout seed live on the device, and one step (53 kernel launches for VTN vc1: 8 per layer;
  round 6: feat_out | prob_out are one projection, the emit kernel also advances the step counter and leaves the NEXT position's
  positional-encoding row, which the input Linear of the next step adds as a residual -- 56 launches before) is
  captured ONCE as a hipGraph and replayed; the host polls the stop flags every `poll` steps.  Several
  utterances decode in lockstep (one row each); rows are independent, so a batch gives the same frames as
  utterance-by-utterance decoding.

Values equal the reference's whenever the prenet dropout is off (the cached K/V of a position are those the
reference recomputes); with dropout on, the reference re-samples the prefix masks every step, which a cache
cannot reproduce -- the per-step masks here are drawn once per generated position.
"""
import math

import torch

from . import modules as Mo
from .ops import functional as Fn
from .ops import kernels as K
from .ops import kernels_decode as KD


def _round_up(n, m):
    return (n + m - 1) // m * m


class ARDecodeSession:
    """Static buffers + captured step graph for one (batch, source-capacity, length-capacity) shape."""

    def __init__(self, model, B, Tcap, Lcap, dtype, device):
        dec = model.decoder
        self.model, self.B, self.Tcap, self.Lcap, self.dtype, self.device = model, B, Tcap, Lcap, dtype, device
        self.r, self.odim = model.decoder_reduction_factor, model.odim
        att0 = dec.decoders[0].self_attn
        self.H, self.dk = att0.h, att0.d_k
        self.D = D = self.H * self.dk
        self.nl = len(dec.decoders)
        self.pre_ln = dec.normalize_before
        self.weights_version = self._version(model)
        cd = lambda t: t.detach().to(dtype).contiguous()            # GEMM operand copies in the compute dtype
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        lin = lambda m: (cd(m.weight), f32(m.bias))
        self.prenet = [lin(blk[0]) for blk in dec.embed[0][0].prenet]
        self.prenet_p = dec.embed[0][0].dropout_rate
        self.embed_lin = lin(dec.embed[0][1])
        pos = dec.embed[1]
        self.pe = pos.table(Lcap, device)
        self.alpha = f32(pos.alpha) if isinstance(pos, Mo.ScaledPositionalEncoding) else None
        self.xscale = 1.0 if self.alpha is not None else pos.xscale
        self.layers = []
        for lyr in dec.decoders:
            sa, ca, ff = lyr.self_attn, lyr.src_attn, lyr.feed_forward
            self.layers.append({
                "w_qkv": cd(torch.cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight], 0)),
                "b_qkv": f32(torch.cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias], 0)),
                "o": lin(sa.linear_out), "q_src": lin(ca.linear_q),
                "w_kv_src": cd(torch.cat([ca.linear_k.weight, ca.linear_v.weight], 0)),
                "b_kv_src": f32(torch.cat([ca.linear_k.bias, ca.linear_v.bias], 0)),
                "o_src": lin(ca.linear_out), "w1": lin(ff.w_1), "w2": lin(ff.w_2),
                "norms": [(f32(n.weight), f32(n.bias), n.eps) for n in (lyr.norm1, lyr.norm2, lyr.norm3)]})
        self.after_norm = (f32(dec.after_norm.weight), f32(dec.after_norm.bias), dec.after_norm.eps) if self.pre_ln else None
        self.feat_out, self.prob_out = lin(model.feat_out), lin(model.prob_out)

        z = lambda *s, dt=dtype: torch.zeros(s, dtype=dt, device=device)
        self.kc = [z(B, Lcap, D) for _ in range(self.nl)]
        self.vc = [z(B, Lcap, D) for _ in range(self.nl)]
        self.src_kv = [z(B, Tcap, 2 * D) for _ in range(self.nl)]
        self.att = z(self.nl, B, self.H, Lcap, Tcap, dt=torch.float32)
        self.outs = z(B, Lcap * self.r, self.odim, dt=torch.float32)
        self.probs = z(B, Lcap * self.r, dt=torch.float32)
        self.prev = z(B, self.odim)
        self.pos = z(1, dt=torch.int32)
        self.stop_at = z(B, dt=torch.int32)
        self.minlen, self.maxlen, self.klen = z(B, dt=torch.int32), z(B, dt=torch.int32), z(B, dt=torch.int32)
        self.threshold = None
        self.graph = None
        # round 6: one packed feat_out | prob_out projection; emit + advance in one launch; the positional row as a residual of the input Linear
        self.w_fp = torch.cat([self.feat_out[0], self.prob_out[0]], 0).contiguous()
        self.b_fp = torch.cat([self.feat_out[1], self.prob_out[1]], 0).contiguous()
        self.ticket = z(1, dt=torch.int32)
        self.pe_cur = z(1, D)                                   # alpha * pe[pos] in the compute dtype, kept current by the emit kernel
        w_e, b_e = self.embed_lin                               # (x W^T + b) * xscale + alpha * pe: xscale rides in the GEMM's alpha and bias
        self.embed_bias = None if b_e is None else (b_e * self.xscale).contiguous()
        self._ll_ok = {}               # (batch, K) -> the fused LayerNorm + projection kernel applies (see _ll)

    @staticmethod
    def _version(model):
        """Identity of the weights a session copied: tensor versions (in-place torch updates, load_state_dict) plus the
        generation counter that optim.FlatAdam advances -- its fused step updates the flat buffer through a raw pointer,
        which bumps no tensor version."""
        return (model.__dict__.get("_s2s_weight_gen", 0),) + tuple(p._version for p in model.parameters())

    # -- one decoder position for all utterances (every launch reads the device-resident `pos`) -------------
    def _lin(self, x, wb, act=None, res=None):
        w, b = wb
        N, Kd = w.shape
        out = torch.empty((x.shape[0], N), dtype=self.dtype, device=self.device)
        return K.gemm(K.operand(x, Kd), K.operand(w, Kd), x.shape[0], N, Kd, out, in_dtype=self.dtype, bias=b, act=act, res=res)

    def _add_norm(self, x, h, norm):
        """(LN(x + h), x + h)"""
        y, s, _, _ = K.layernorm_fwd(h, norm[0], norm[1], norm[2], res=x, need_stats=False)
        return y, s

    # -- one decoder position: 8 launches per layer (the LayerNorms ride in the projections that consume them, the residual
    #    adds in the projections that produce the sublayer outputs) ---------------------------------------------------------
    def _attend_self(self, i, qkv):
        B, H, dk, D = self.B, self.H, self.dk, self.D
        ctx = torch.empty((B, D), dtype=self.dtype, device=self.device)
        KD.decode_attn(qkv, 0, 3 * D, self.kc[i], 0, self.vc[i], 0, D, self.Lcap * D, qkv, D, 2 * D, 3 * D, self.pos, None,
                       self.Lcap, 1.0 / math.sqrt(dk), ctx, B, H, dk)
        return ctx

    def _attend_src(self, i, q):
        B, H, dk, D = self.B, self.H, self.dk, self.D
        ctx = torch.empty((B, D), dtype=self.dtype, device=self.device)
        a = self.att[i]
        KD.decode_attn(q, 0, D, self.src_kv[i], 0, self.src_kv[i], D, 2 * D, self.Tcap * 2 * D, None, 0, 0, 0, self.pos, self.klen,
                       self.Tcap, 1.0 / math.sqrt(dk), ctx, B, H, dk, att=a, att_strides=(a.stride(0), a.stride(1), a.stride(2)))
        return ctx

    def _ll(self, x, w, bias, *, norm=None, act=None, res=None, y_out=None, drop_p=0.0, seed=(None, 0)):
        """act(LN(x) . w^T + bias) [dropout] (+ res): the fused skinny kernel where it applies (batch <= 64, K in whole 16-byte
        vectors and short enough for its register-resident form: decided per (batch, K) once, before the step is captured),
        else LayerNorm kernel + GEMM with the same epilogue (same values: the fused kernel normalises exactly as norm.hip)."""
        key = (x.shape[0], x.shape[1])
        ok = self._ll_ok.get(key)
        if ok is None:
            ok = self._ll_ok[key] = KD.ln_linear_supported(self.dtype, x.shape[0], x.shape[1])
        if ok:
            return KD.ln_linear(x, w, bias, norm=norm, act=act, res=res, y_out=y_out, drop_p=drop_p, seed=seed)
        y = x
        if norm is not None:
            y, _, _, _ = K.layernorm_fwd(x, norm[0], norm[1], norm[2], need_stats=False)
            if y_out is not None:
                K.cast(y, self.dtype, out=y_out)         # a kernel, not a memcpy node (the step is captured)
        N, Kd = w.shape
        out = torch.empty((x.shape[0], N), dtype=self.dtype, device=self.device)
        return K.gemm(K.operand(y, Kd), K.operand(w, Kd), x.shape[0], N, Kd, out, in_dtype=self.dtype, bias=bias, act=act, res=res,
                      drop_p=drop_p, seed=seed)

    def _step(self):
        ll, lin = self._ll, self._lin            # LayerNorm / dropout fused into the projection | plain skinny projection
        x = self.prev
        for wb in self.prenet:                                  # Linear-ReLU-dropout in one launch, dropout ALWAYS on (F9)
            if self.prenet_p > 0.0:
                x = ll(x, wb[0], wb[1], act="relu", drop_p=self.prenet_p, seed=K.new_seed(self.device))
            else:
                x = lin(x, wb, act="relu")
        w_e = self.embed_lin[0]
        x = K.gemm(K.operand(x, w_e.shape[1]), K.operand(w_e, w_e.shape[1]), self.B, w_e.shape[0], w_e.shape[1],
                   torch.empty((self.B, w_e.shape[0]), dtype=self.dtype, device=self.device), in_dtype=self.dtype, bias=self.embed_bias,
                   alpha=self.xscale, res=self.pe_cur, ldr=0)     # + alpha * pe[pos] (embedding.py:115-125): the row the last emit left
        new = lambda: torch.empty((self.B, self.D), dtype=self.dtype, device=self.device)
        if self.pre_ln:       # decoder_layer.py:85-132 with normalize_before: x += f(LN(x)); x is the residual stream
            for i, L in enumerate(self.layers):
                n1, n2, n3 = L["norms"]
                ctx = self._attend_self(i, ll(x, L["w_qkv"], L["b_qkv"], norm=n1))
                x = lin(ctx, L["o"], res=x)
                ctx = self._attend_src(i, ll(x, L["q_src"][0], L["q_src"][1], norm=n2))
                x = lin(ctx, L["o_src"], res=x)
                h = ll(x, L["w1"][0], L["w1"][1], norm=n3, act="relu")
                x = lin(h, L["w2"], res=x)
            last, norm = x, self.after_norm
        else:                 # post-norm: x = LN(x + f(x)); s = the sum waiting for its LayerNorm, applied by its consumer
            s, norm = None, None
            for i, L in enumerate(self.layers):
                n1, n2, n3 = L["norms"]
                if s is None:
                    qkv = lin(x, (L["w_qkv"], L["b_qkv"]))
                else:
                    x = new()
                    qkv = ll(s, L["w_qkv"], L["b_qkv"], norm=norm, y_out=x)
                s = lin(self._attend_self(i, qkv), L["o"], res=x)
                x = new()
                q = ll(s, L["q_src"][0], L["q_src"][1], norm=n1, y_out=x)
                s = lin(self._attend_src(i, q), L["o_src"], res=x)
                x = new()
                h = ll(s, L["w1"][0], L["w1"][1], norm=n2, act="relu", y_out=x)
                s = lin(h, L["w2"], res=x)
                norm = n3
            last = s
        out = ll(last, self.w_fp, self.b_fp, norm=norm)
        KD.decode_emit_advance(out, self.r, self.odim, self.threshold, self.minlen, self.maxlen, self.pos, self.outs, self.probs, self.prev,
                               self.stop_at, K.SEED.tensor(self.device).data_ptr(), 0x10001, self.ticket, pe=self.pe, alpha=self.alpha,
                               pe_next=self.pe_cur)

    def _reset(self):
        self.pos.zero_()
        self.stop_at.zero_()
        self.prev.zero_()
        self.ticket.zero_()
        a = self.alpha if self.alpha is not None else 1.0
        self.pe_cur.copy_((self.pe[0:1] * a).to(self.dtype))       # position 0 (the emit kernel writes the rows of the later ones)

    def _capture(self):
        """Warm up eagerly on a side stream (lazy initialisation), then capture one step."""
        K.reset_op_counter()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            K.reset_op_counter()
            self._step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        K.reset_op_counter()
        with torch.cuda.graph(g):
            self._step()
        self.graph = g
        self._reset()

    # -- whole utterances --------------------------------------------------------------------------------------
    def run(self, hs, hlens, threshold, minlenratio, maxlenratio, poll=16, use_graph=True):
        """hs (B, Tenc, D) encoder memory, hlens list of valid memory lengths.  Returns per utterance
        (frames (L*r, odim) fp32 before the postnet, probs (L*r,), att_ws (layers, H, L, Tenc_b)) and L."""
        B, Tenc, D = hs.shape
        assert B == self.B and Tenc <= self.Tcap and D == self.D
        r = self.r
        maxlen = [int(t * maxlenratio / r) for t in hlens]        # vtn.py:334-335, per utterance
        minlen = [int(t * minlenratio / r) for t in hlens]
        steps_cap = max(1, max(max(a, b) for a, b in zip(maxlen, minlen)))
        if steps_cap > self.Lcap:
            raise ValueError(f"decode length {steps_cap} exceeds the session capacity {self.Lcap}")
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=self.device)
        self.minlen.copy_(i32(minlen))
        self.maxlen.copy_(i32(maxlen))
        self.klen.copy_(i32(list(hlens)))
        for i, L in enumerate(self.layers):                         # source K/V: projected ONCE (reference: every step)
            kv = torch.empty((B, Tenc, 2 * D), dtype=self.dtype, device=self.device)
            K.gemm(K.operand(hs, D), K.operand(L["w_kv_src"], D), B * Tenc, 2 * D, D, kv, in_dtype=self.dtype, bias=L["b_kv_src"])
            self.src_kv[i][:, :Tenc].copy_(kv)
        if self.threshold != float(threshold):
            self.threshold, self.graph = float(threshold), None      # the threshold is a captured kernel argument
        self._reset()
        if use_graph and self.graph is None:
            self._capture()
        done = 0
        while True:
            n = min(poll, steps_cap - done)
            for _ in range(n):
                if use_graph:
                    self.graph.replay()
                else:
                    K.reset_op_counter()
                    self._step()
            done += n
            stop = self.stop_at.tolist()                             # the only host sync: once per `poll` steps
            if all(s > 0 for s in stop) or done >= steps_cap:
                break
        res = []
        for b in range(B):
            Lb = stop[b] if stop[b] > 0 else done
            res.append((self.outs[b, :Lb * r], self.probs[b, :Lb * r], self.att[:, b, :, :Lb, :hlens[b]], Lb))
        return res


MAX_SESSION_BATCH = 64


def decode(model, hs, hlens, inference_args, poll=16, use_graph=True):
    """Batched generation for an AR model (VTN / TransformerTTS): hs (B, Tenc, D), hlens host ints.
    Returns [(outs (L, odim), probs (L,), att_ws (layers, H, L/r, Tenc_b)), ...] exactly as the reference's
    `inference` returns for each utterance (vtn.py:391-394)."""
    B, Tenc, _ = hs.shape
    if B > MAX_SESSION_BATCH:          # the decode-step kernels hold a batch of at most 64 rows per workgroup: larger batches in groups
        out = []
        for b0 in range(0, B, MAX_SESSION_BATCH):
            out += decode(model, hs[b0:b0 + MAX_SESSION_BATCH], list(hlens)[b0:b0 + MAX_SESSION_BATCH], inference_args, poll, use_graph)
        return out
    dtype, device = hs.dtype, hs.device
    r = model.decoder_reduction_factor
    need_L = max(1, max(max(int(t * inference_args["maxlenratio"] / r), int(t * inference_args["minlenratio"] / r)) for t in hlens))
    cache = model.__dict__.setdefault("_decode_sessions", {})
    key = (B, dtype, device)
    sess = cache.get(key)
    if (sess is None or sess.Tcap < Tenc or sess.Lcap < need_L or sess.weights_version != ARDecodeSession._version(model)):
        Tcap = _round_up(max(Tenc, sess.Tcap if sess else 0), 64)
        Lcap = _round_up(max(need_L, sess.Lcap if sess else 0), 64)
        sess = ARDecodeSession(model, B, Tcap, Lcap, dtype, device)
        cache[key] = sess
    rows = sess.run(hs.contiguous(), list(hlens), inference_args["threshold"], inference_args["minlenratio"],
                    inference_args["maxlenratio"], poll=poll, use_graph=use_graph)
    # postnet per group of equal length (Conv1d zero-pads at the true utterance end; BatchNorm uses running stats)
    out = [None] * B
    groups = {}
    for b, (_, _, _, Lb) in enumerate(rows):
        groups.setdefault(Lb, []).append(b)
    for Lb, idx in groups.items():
        before = Fn.to_compute(torch.stack([rows[b][0] for b in idx]))          # (n, L*r, odim)
        after = before
        if model.postnet is not None:
            after = Fn.add_dropout(before, model.postnet(before.contiguous()), 0.0)
        for j, b in enumerate(idx):
            out[b] = (after[j].float(), rows[b][1].clone(), rows[b][2].clone())
    return out
