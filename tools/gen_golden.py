"""Generate tests/golden/*.npz by IMPORTING the reference (unilight/seq2seq-vc) in this container.

Runs only where /root/reference exists (never on the GPU box).  Each fixture holds: the model
config, the full state_dict, the inputs, every output of the reference's forward, selected
intermediates (attention maps, log_p_attn, MAS paths, durations), the scalar losses and the
gradients of all parameters -- so both the CPU oracle (tests/test_oracle_golden.py) and the HIP
path (tests/test_gpu_models.py) are pinned against the reference itself.

Determinism: every nn.Dropout p and Prenet/Postnet dropout is forced to 0, torch.manual_seed before
construction and forward, and the two torch.randn draws of the stochastic duration predictor are
captured and stored.

    python tools/gen_golden.py            # writes tests/golden/*.npz and prints oracle-vs-reference errors
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("S2SVC_REFERENCE", "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def import_reference():
    nb = types.ModuleType("numba")

    class _T:
        def __getitem__(self, k):
            return self

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], (tuple, _T)):
            return a[0]
        return lambda f: f

    nb.jit = nb.njit = jit
    nb.float64 = nb.float32 = nb.int8 = nb.int64 = nb.boolean = _T()
    sys.modules["numba"] = nb
    sys.modules["seq2seq_vc.losses.diffsinger_l2_loss"] = types.ModuleType("x")
    sys.path.insert(0, REF)
    import seq2seq_vc.losses as L
    import seq2seq_vc.models as M
    import seq2seq_vc.modules.alignments as A
    return M, L, A


def kill_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout_rate"):
            m.dropout_rate = 0.0


def to_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def pack(d, prefix, sd_like):
    for k, v in sd_like.items():
        d[prefix + k] = to_np(v)


def save(name, cfg, arrays):
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    arrays = dict(arrays)
    arrays["__cfg__"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(out, **arrays)
    print(f"wrote {out}  ({os.path.getsize(out) / 1024:.0f} KiB, {len(arrays)} arrays)")


def synth_batch(B, Tmax_in, Tmax_out, idim, odim, seed, min_frac=0.6, multiple=1):
    g = torch.Generator().manual_seed(seed)
    ilens = torch.randint(int(Tmax_in * min_frac), Tmax_in + 1, (B,), generator=g)
    olens = torch.randint(int(Tmax_out * min_frac), Tmax_out + 1, (B,), generator=g)
    ilens[0], olens[0] = Tmax_in, Tmax_out
    xs = torch.randn(B, Tmax_in, idim, generator=g)
    ys = torch.randn(B, Tmax_out, odim, generator=g)
    for b in range(B):
        xs[b, ilens[b]:] = 0
        ys[b, olens[b]:] = 0
    labels = torch.zeros(B, Tmax_out)
    for b in range(B):
        labels[b, olens[b] - 1:] = 1.0
    return xs, ilens, ys, labels, olens


def maxerr(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    same_inf = torch.isinf(a) & torch.isinf(b) & (a == b)
    return float(torch.where(same_inf, torch.zeros_like(a), (a - b).abs()).max())


# ------------------------------------------------------------------------------------------------
def gen_vtn(M, L, name, cfg, B, Ti, To, seed, train=True):
    from oracle import models as OM
    torch.manual_seed(seed)
    model = M.VTN(**cfg)
    kill_dropout(model)
    model.train(train)
    xs, ilens, ys, labels, olens = synth_batch(B, Ti, To, cfg["idim"], cfg["odim"], seed + 1)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    after, before, logits, ys_, labels_, olens_, (att_ws, ilens_ds, olens_in) = model(xs, ilens, ys, labels, olens)
    crit = L.Seq2SeqLoss(bce_pos_weight=10.0)
    l1, bce = crit(after, before, logits, ys_, labels_, olens_)
    ga = L.GuidedMultiHeadAttentionLoss(sigma=0.4, alpha=1.0)(att_ws[0], ilens_ds, olens_in)
    model.zero_grad()
    (l1 + bce).backward()
    arr = {}
    pack(arr, "sd.", sd0)
    pack(arr, "sd_after.", {k: v for k, v in model.state_dict().items() if "running" in k or "num_batches" in k})
    pack(arr, "grad.", {k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    arr.update({"in.xs": to_np(xs), "in.ilens": to_np(ilens), "in.ys": to_np(ys), "in.labels": to_np(labels), "in.olens": to_np(olens),
                "out.after": to_np(after), "out.before": to_np(before), "out.logits": to_np(logits), "out.ys": to_np(ys_),
                "out.labels": to_np(labels_), "out.olens": to_np(olens_), "out.ilens_ds": to_np(ilens_ds), "out.olens_in": to_np(olens_in),
                "loss.l1": to_np(l1), "loss.bce": to_np(bce), "loss.guided_attn": to_np(ga)})
    for i, a in enumerate(att_ws):
        arr[f"out.att_ws.{i}"] = to_np(a)
    nl = len(model.encoder.encoders)
    for i in range(nl):
        arr[f"attn.encoder.encoders.{i}.self_attn"] = to_np(model.encoder.encoders[i].self_attn.attn)
    save(name, dict(cfg, __train__=train, __model__="VTN"), arr)
    # oracle check
    sdo = {k: v.clone() for k, v in sd0.items()}
    o = OM.vtn_forward(sdo, cfg, xs, ilens, ys, labels, olens, training=train)
    print(f"  oracle-vs-ref {name}: after {maxerr(o[0], after):.2e} before {maxerr(o[1], before):.2e} logits {maxerr(o[2], logits):.2e} "
          f"att {maxerr(o[6][0][0], att_ws[0]):.2e}")
    l1o, bceo = OM.seq2seq_loss(o[0], o[1], o[2], o[3], o[4], o[5])
    gao = OM.guided_attention_loss(o[6][0][0], o[6][1], o[6][2])
    print(f"  losses: l1 {maxerr(l1o, l1):.2e} bce {maxerr(bceo, bce):.2e} ga {maxerr(gao, ga):.2e}")
    return model


def gen_vtn_inference(M, name, cfg, T, seed, maxlenratio=2.0, fire=False, minlenratio=0.0, tts=False):
    """AR generation fixture.  fire=True: the stop threshold is placed inside the range of the stop probabilities the
    reference itself produces, so generation ends through the threshold test (vtn.py:378-381) and not at maxlen."""
    from oracle import models as OM
    torch.manual_seed(seed)
    model = (M.TransformerTTS if tts else M.VTN)(**cfg)
    kill_dropout(model)
    model.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randint(1, cfg["idim"] - 1, (T,), generator=g) if tts else torch.randn(T, cfg["idim"], generator=g)
    args = {"threshold": 2.0, "minlenratio": minlenratio, "maxlenratio": maxlenratio}
    with torch.no_grad():
        outs, probs, att_ws = model.inference(x, args)
        if fire:
            r = cfg["decoder_reduction_factor"]
            per_step = probs.view(-1, r).max(dim=1).values            # a step stops when any of its r probs fires
            n = per_step.numel()
            k = max(2, n // 2)                                         # aim for the middle of the utterance
            thr = float(per_step[k]) * (1 - 1e-3)                      # clear margin: fp32 noise cannot flip the step
            first = int((per_step >= thr).nonzero()[0])
            assert (per_step[:first] < thr * (1 - 1e-3)).all() or first == 0
            args = dict(args, threshold=thr)
            outs, probs, att_ws = model.inference(x, args)
            print(f"  {name}: threshold {thr:.6f} fires at step {first + 1} of {n}; generated {outs.shape[0]} frames")
    arr = {}
    pack(arr, "sd.", model.state_dict())
    arr.update({"in.x": to_np(x), "out.outs": to_np(outs), "out.probs": to_np(probs), "out.att_ws": to_np(att_ws)})
    save(name, dict(cfg, __model__="TransformerTTS" if tts else "VTN", __inference__=args), arr)
    with torch.no_grad():
        o = OM.vtn_inference({k: v.clone() for k, v in model.state_dict().items()}, cfg, x, tts=tts, **args)
    print(f"  oracle-vs-ref {name}: outs {maxerr(o[0], outs):.2e} probs {maxerr(o[1], probs):.2e} att {maxerr(o[2], att_ws):.2e}")


def gen_tts(M, L, name, cfg, B, Ti, To, seed):
    from oracle import models as OM
    torch.manual_seed(seed)
    model = M.TransformerTTS(**cfg)
    kill_dropout(model)
    model.train()
    g = torch.Generator().manual_seed(seed + 1)
    ilens = torch.randint(Ti // 2, Ti + 1, (B,), generator=g)
    ilens[0] = Ti
    xs = torch.randint(1, cfg["idim"] - 1, (B, Ti), generator=g)
    for b in range(B):
        xs[b, ilens[b]:] = 0
    _, _, ys, labels, olens = synth_batch(B, Ti, To, 1, cfg["odim"], seed + 2)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    after, before, logits, ys_, labels_, olens_, (att_ws, ilens2, olens_in) = model(xs.clone(), ilens, ys, labels, olens)
    l1, bce = L.Seq2SeqLoss(bce_pos_weight=10.0)(after, before, logits, ys_, labels_, olens_)
    model.zero_grad()
    (l1 + bce).backward()
    arr = {}
    pack(arr, "sd.", sd0)
    pack(arr, "grad.", {k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    arr.update({"in.xs": to_np(xs), "in.ilens": to_np(ilens), "in.ys": to_np(ys), "in.labels": to_np(labels), "in.olens": to_np(olens),
                "out.after": to_np(after), "out.before": to_np(before), "out.logits": to_np(logits), "out.ys": to_np(ys_),
                "out.labels": to_np(labels_), "out.olens": to_np(olens_), "loss.l1": to_np(l1), "loss.bce": to_np(bce)})
    save(name, dict(cfg, __model__="TransformerTTS"), arr)
    o = OM.tts_forward({k: v.clone() for k, v in sd0.items()}, cfg, xs.clone(), ilens, ys, labels, olens)
    print(f"  oracle-vs-ref {name}: after {maxerr(o[0], after):.2e} before {maxerr(o[1], before):.2e} logits {maxerr(o[2], logits):.2e}")


def gen_aasvc(M, L, A, name, cfg, B, Ti, To, seed, lambda_align=2.0):
    from oracle import models as OM
    torch.manual_seed(seed)
    model = M.AASVC(**cfg)
    kill_dropout(model)
    model.train()
    xs, ilens, ys, _, olens = synth_batch(B, Ti, To, cfg["idim"], cfg["odim"], seed + 1, min_frac=0.75)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    drawn = []
    real_randn = torch.randn

    def spy(*a, **k):
        t = real_randn(*a, **k)
        drawn.append(t.clone())
        return t

    torch.randn = spy
    try:
        torch.manual_seed(seed + 5)
        ret = model(xs, ilens, ys, olens, xs, dp_lengths=ilens)
    finally:
        torch.randn = real_randn
    l1 = L.L1Loss()(ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
    fs = L.ForwardSumLoss()(ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
    total = l1 + lambda_align * (fs + ret["bin_loss"])
    if "dur_nll" in ret:
        total = total + torch.sum(ret["dur_nll"].float())
    model.zero_grad()
    total.backward()
    arr = {}
    pack(arr, "sd.", sd0)
    pack(arr, "grad.", {k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    arr.update({"in.xs": to_np(xs), "in.ilens": to_np(ilens), "in.ys": to_np(ys), "in.olens": to_np(olens),
                "out.after": to_np(ret["after_outs"]), "out.before": to_np(ret["before_outs"]), "out.ds": to_np(ret["ds"]),
                "out.ilens": to_np(ret["ilens"]), "out.bin_loss": to_np(ret["bin_loss"]), "out.log_p_attn": to_np(ret["log_p_attn"]),
                "out.olens_reduced": to_np(ret["olens_reduced"]), "out.olens": to_np(ret["olens"]), "out.ys": to_np(ret["ys"]),
                "loss.l1": to_np(l1), "loss.forward_sum": to_np(fs), "loss.total": to_np(total)})
    if "dur_nll" in ret:
        arr["out.dur_nll"] = to_np(ret["dur_nll"])
        arr["in.sdp_noise"] = to_np(drawn[0])
    else:
        arr["out.d_outs"] = to_np(ret["d_outs"])
    # MAS paths + decision margins straight from the reference's own functions
    lp = ret["log_p_attn"].detach()
    margins = []
    for b in range(B):
        cur = lp[b, : int(ret["olens_reduced"][b]), : int(ret["ilens"][b])].numpy()
        path = A._monotonic_alignment_search(cur)
        arr[f"out.mas_path.{b}"] = np.asarray(path)
    save(name, dict(cfg, __model__="AASVC", __lambda_align__=lambda_align), arr)
    noise = drawn[0] if drawn else None
    o = OM.aasvc_forward({k: v.clone() for k, v in sd0.items()}, cfg, xs, ilens, ys, olens, dp_inputs=xs, noise=noise)
    print(f"  oracle-vs-ref {name}: after {maxerr(o['after_outs'], ret['after_outs']):.2e} log_p {maxerr(o['log_p_attn'], ret['log_p_attn']):.2e} "
          f"ds_equal {bool(torch.equal(o['ds'], ret['ds']))} bin {maxerr(o['bin_loss'], ret['bin_loss']):.2e} margin {o['mas_margin']:.2e}")
    if "dur_nll" in ret:
        print(f"  dur_nll {maxerr(o['dur_nll'], ret['dur_nll']):.2e}")
    fso = OM.forward_sum_loss(o["log_p_attn"], o["ilens"], o["olens_reduced"])
    print(f"  forward_sum {maxerr(fso, fs):.2e} l1 {maxerr(OM.l1_loss(o['after_outs'], o['before_outs'], o['ys'], o['olens']), l1):.2e}")


def gen_aasvc_inference(M, name, cfg, T, seed, To=None):
    """AASVC.inference (models/aas_vc.py:531-603, the non-teacher-forced branch): without a target (the decode path of
    bin/vc_decode.py) and with one (the debug path that also returns ds / log_p_attn / ilens).  eval() mode; the randn
    draw of the stochastic duration predictor's inverse pass is captured."""
    from oracle import models as OM
    torch.manual_seed(seed)
    model = M.AASVC(**cfg)
    # default init leaves the duration heads near zero (all durations 1); scale the duration predictor so that the
    # predicted durations spread over 0..MAX_DP_OUTPUT and the clamp / zero-duration handling is exercised
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.startswith("duration_predictor") and p.dim() > 1:
                p.mul_(3.0)
    kill_dropout(model)
    model.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(T, cfg["idim"], generator=g)
    y = torch.randn(To, cfg["odim"], generator=g) if To else None
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    drawn = []
    real_randn = torch.randn

    def spy(*a, **k):
        t = real_randn(*a, **k)
        drawn.append(t.clone())
        return t

    torch.randn = spy
    try:
        torch.manual_seed(seed + 5)
        with torch.no_grad():
            out = model.inference(x, tgt_speech=y, dp_input=x)
    finally:
        torch.randn = real_randn
    arr = {}
    pack(arr, "sd.", sd0)
    arr.update({"in.x": to_np(x), "out.outs": to_np(out[0]), "out.d_outs": to_np(out[1])})
    if y is not None:
        arr.update({"in.y": to_np(y), "out.ds": to_np(out[2]), "out.log_p_attn": to_np(out[3]), "out.ilens": to_np(out[4])})
    if drawn:
        arr["in.sdp_noise"] = to_np(drawn[0])
    save(name, dict(cfg, __model__="AASVC", __train__=False), arr)
    noise = drawn[0] if drawn else None
    ilens = torch.tensor([T])
    o = OM.aasvc_forward({k: v.clone() for k, v in sd0.items()}, cfg, x[None], ilens, None if y is None else y[None],
                         None if y is None else torch.tensor([To]), dp_inputs=x[None], noise=noise, training=False, inference=True)
    print(f"  oracle-vs-ref {name}: outs {maxerr(o['after_outs'][0], out[0]):.2e} d_outs equal {bool(torch.equal(o['d_outs'][0].float(), out[1].float()))} "
          f"(durations {out[1].flatten().tolist()})")


def _fw_reference_layer(c):
    """The reference's own layer object for a tests/fullwidth.py case."""
    from seq2seq_vc.layers.positional_encoding import RelPositionalEncoding
    from seq2seq_vc.modules.conformer.convolution import ConvolutionModule
    from seq2seq_vc.modules.conformer.encoder_layer import EncoderLayer as ConformerLayer
    from seq2seq_vc.modules.conformer.swish import Swish
    from seq2seq_vc.modules.transformer.attention import MultiHeadedAttention, RelPositionMultiHeadedAttention
    from seq2seq_vc.modules.transformer.decoder_layer import DecoderLayer
    from seq2seq_vc.modules.transformer.encoder_layer import EncoderLayer
    from seq2seq_vc.modules.transformer.positionwise_feed_forward import PositionwiseFeedForward
    d, h, u = c["d"], c["h"], c["units"]
    if c["kind"] == "encoder":
        return EncoderLayer(d, MultiHeadedAttention(h, d, 0.0), PositionwiseFeedForward(d, u, 0.0), 0.0, c["pre_ln"], False), None
    if c["kind"] == "decoder":
        return DecoderLayer(d, MultiHeadedAttention(h, d, 0.0), MultiHeadedAttention(h, d, 0.0), PositionwiseFeedForward(d, u, 0.0),
                            0.0, c["pre_ln"], False), None
    layer = ConformerLayer(d, RelPositionMultiHeadedAttention(h, d, 0.0, False), PositionwiseFeedForward(d, u, 0.0, Swish()),
                           PositionwiseFeedForward(d, u, 0.0, Swish()), ConvolutionModule(d, c["k"], Swish()), 0.0, c["pre_ln"], False)
    return layer, RelPositionalEncoding(d, 0.0)


def gen_fullwidth(name):
    """Full-width single layers (tests/fullwidth.py): reference layer, seeded weights, output + gradients."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullwidth as FW
    from oracle import nets as N
    c = FW.CASES[name]
    layer, posenc = _fw_reference_layer(c)
    layer.train()                       # BatchNorm of the Conformer layer uses batch statistics, as in training
    shapes = [(k, tuple(p.shape)) for k, p in layer.named_parameters()]
    state = FW.seeded_state(shapes, c["seed"])
    with torch.no_grad():
        for k, p in layer.named_parameters():
            p.copy_(state[k])
    x, mem, dy = FW.inputs(c)
    x.requires_grad_(True)
    lens = torch.tensor(c["lens"])
    T = c["T"]
    key_mask = (torch.arange(T)[None, :] < lens[:, None]).unsqueeze(1)          # (B,1,T) True = valid
    if c["kind"] == "encoder":
        out, _ = layer(x, key_mask)
    elif c["kind"] == "decoder":
        mem.requires_grad_(True)
        mlens = torch.tensor(c["mlens"])
        mem_mask = (torch.arange(c["Tm"])[None, :] < mlens[:, None]).unsqueeze(1)
        tgt_mask = key_mask & torch.tril(torch.ones(T, T, dtype=torch.bool))[None]
        out, _, _, _ = layer(x, tgt_mask, mem, mem_mask)
    else:
        xs, pos_emb = posenc(x)          # (x * sqrt(d), pos_emb (1, 2T-1, d)); the layer input is the scaled x
        (out, _), _ = layer((xs, pos_emb), key_mask)
    (out * dy).sum().backward()
    arr = {"out": to_np(out), "dx": to_np(x.grad).reshape(-1)[::3].copy(),
           "chk.x": np.int64(FW.checksum(x)), "chk.dy": np.int64(FW.checksum(dy))}
    if mem is not None:
        arr["dmem"] = to_np(mem.grad).reshape(-1)[::3].copy()
    for k, p in layer.named_parameters():
        arr["chk.w." + k] = np.int64(FW.checksum(state[k]))
        arr["grad." + k] = to_np(p.grad).reshape(-1)[::FW.grad_stride(p.numel())].copy()
    for k, b in layer.named_buffers():
        arr["buf." + k] = to_np(b)
    save(name, dict({k: v for k, v in c.items()}, __model__="layer"), arr)
    # the oracle's restatement of the same layer
    from oracle.nets import P, Runtime
    sd = {k: v.clone() for k, v in state.items()}
    for k, b in _fw_reference_layer(c)[0].named_buffers():
        sd[k] = b.clone()
    rt = Runtime(True, False)
    with torch.no_grad():
        xd = x.detach()
        if c["kind"] == "encoder":
            o = N.encoder_layer(P(sd, ""), xd, key_mask, c["h"], rt, 0.0, 0.0, c["pre_ln"], "l")
        elif c["kind"] == "decoder":
            o = N.decoder_layer(P(sd, ""), xd, tgt_mask, mem.detach(), mem_mask, c["h"], rt, 0.0, c["pre_ln"], "l")
        else:
            xs, pe = N.rel_posenc(xd, rt, 0.0)
            o = N.conformer_layer(P(sd, ""), xs, pe, key_mask, c["h"], rt, 0.0, 0.0, c["pre_ln"], False, "l")
    print(f"  oracle-vs-ref {name}: out {maxerr(o, out):.2e} (|out| max {float(out.abs().max()):.2f})")


def gen_fs2vc(M, L, name, cfg, B, Ti, seed):
    """FastSpeechVC (models/fastspeech_vc.py:244-466) with teacher durations: training forward + L1 + duration loss +
    gradients, the inference path, and DurationCalculator (utils/duration_calculator.py) known answers."""
    from oracle import models as OM
    torch.manual_seed(seed)
    model = M.FastSpeechVC(**cfg)
    kill_dropout(model)
    model.train()
    g = torch.Generator().manual_seed(seed + 1)
    ilens = torch.randint(int(Ti * 0.7), Ti + 1, (B,), generator=g)
    ilens[0] = Ti
    xs = torch.randn(B, Ti, cfg["idim"], generator=g)
    xs[torch.arange(Ti)[None] >= ilens[:, None]] = 0.0
    sub = cfg.get("encoder_input_layer") == "conv2d" or cfg.get("encoder_type", "transformer") == "transformer"
    tlens = (((ilens - 1) // 2 - 1) // 2) if sub else ilens
    Tx = int(tlens.max())
    ds = torch.randint(0, 5, (B, Tx), generator=g)
    ds[:, 0] = 2
    ds[torch.arange(Tx)[None] >= tlens[:, None]] = 0
    r = cfg.get("teacher_model_decoder_reduction_factor", 4)
    olens = ds.sum(1) * r
    ys = torch.randn(B, int(olens.max()), cfg["odim"], generator=g)
    ys[torch.arange(ys.shape[1])[None] >= olens[:, None]] = 0.0
    dlens = torch.full((B,), Tx)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    before, after, d_outs, ilens_, olens_, ys_ = model(xs, ilens, ys, olens, ds, dlens, xs, dp_lengths=ilens)
    l1 = L.L1Loss()(after, before, ys_, olens_)
    dl = L.DurationPredictorLoss()(d_outs, ds, ilens_)
    model.zero_grad()
    (l1 + dl).backward()
    arr = {}
    pack(arr, "sd.", sd0)
    pack(arr, "grad.", {k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    pack(arr, "sd_after.", {k: v for k, v in model.state_dict().items() if "running" in k})
    arr.update({"in.xs": to_np(xs), "in.ilens": to_np(ilens), "in.ys": to_np(ys), "in.olens": to_np(olens), "in.ds": to_np(ds),
                "in.dlens": to_np(dlens), "out.before": to_np(before), "out.after": to_np(after), "out.d_outs": to_np(d_outs),
                "out.ilens": to_np(ilens_), "out.olens": to_np(olens_), "out.ys": to_np(ys_), "loss.l1": to_np(l1), "loss.duration": to_np(dl)})
    model.eval()
    with torch.no_grad():
        x1 = xs[1, : int(ilens[1])]
        outs, d1 = model.inference(x1, dp_input=x1)
    arr.update({"inf.x": to_np(x1), "inf.outs": to_np(outs), "inf.d_outs": to_np(d1)})
    # DurationCalculator known answers (4-D transformer case and 2-D case)
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_duration_calculator", os.path.join(REF, "seq2seq_vc", "utils", "duration_calculator.py"))
    dc_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc_mod)
    dc = dc_mod.DurationCalculator()
    att4 = torch.softmax(torch.randn(2, 3, 37, 11, generator=g) * 3, dim=-1)
    att2 = torch.softmax(torch.randn(29, 9, generator=g) * 2, dim=-1)
    d4, f4 = dc(att4)
    d2, f2 = dc(att2)
    arr.update({"dc.att4": to_np(att4), "dc.dur4": to_np(d4), "dc.focus4": to_np(f4), "dc.att2": to_np(att2), "dc.dur2": to_np(d2),
                "dc.focus2": to_np(f2)})
    save(name, dict(cfg, __model__="FastSpeechVC"), arr)
    o = OM.fastspeech_vc_forward({k: v.clone() for k, v in sd0.items()}, cfg, xs, ilens, ys, olens, ds, dp_inputs=xs)
    print(f"  oracle-vs-ref {name}: after {maxerr(o[1], after):.2e} d_outs {maxerr(o[2], d_outs):.2e} ilens {bool(torch.equal(o[3], ilens_))}")
    with torch.no_grad():
        oi = OM.fastspeech_vc_forward({k: v.clone() for k, v in model.state_dict().items()}, cfg, x1[None], torch.tensor([x1.shape[0]]),
                                      dp_inputs=x1[None], training=False, inference=True)
    print(f"  inference: outs {maxerr(oi[1][0], outs):.2e} durations equal {bool(torch.equal(oi[2][0], d1))} {d1.tolist()}")
    od4, of4 = OM.duration_calculator(att4)
    print(f"  duration calculator: {bool(torch.equal(od4, d4))} focus {maxerr(of4, f4):.2e}")


def gen_mas_kats(A):
    """Known-answer vectors for the alignment search, from the reference's own code."""
    arr = {}
    k1 = np.log(np.array([[.7, .2, .1], [.6, .3, .1], [.2, .6, .2], [.1, .6, .3], [.1, .2, .7], [.05, .15, .8]], dtype=np.float32))
    k2 = np.full((6, 3), np.log(1 / 3), dtype=np.float32)
    k3 = torch.log_softmax(torch.from_numpy(np.random.default_rng(0).standard_normal((3, 5)).astype(np.float32)), -1).numpy()
    for i, k in enumerate((k1, k2, k3), 1):
        arr[f"kat{i}.logp"] = k
        arr[f"kat{i}.path"] = np.asarray(A._monotonic_alignment_search(k))
    lp = torch.log_softmax(torch.randn(2, 6, 3, generator=torch.Generator().manual_seed(7)), dim=-1)
    ds, bl = A.viterbi_decode(lp, torch.tensor([3, 2]), torch.tensor([6, 4]))
    arr.update({"kat4.logp": to_np(lp), "kat4.text_lens": np.array([3, 2]), "kat4.feat_lens": np.array([6, 4]), "kat4.ds": to_np(ds),
                "kat4.bin_loss": to_np(bl)})
    # a batch of random cases with paths
    g = torch.Generator().manual_seed(11)
    for i, (Tf, Tx) in enumerate([(40, 12), (64, 16), (30, 30), (25, 40), (200, 64)]):
        lp = torch.log_softmax(torch.randn(Tf, Tx, generator=g) * 2, dim=-1).numpy()
        arr[f"rand{i}.logp"] = lp
        arr[f"rand{i}.path"] = np.asarray(A._monotonic_alignment_search(lp))
    save("mas_kats", {"__model__": "MAS"}, arr)
    from oracle import mas as omas
    for k in [k for k in arr if k.endswith(".logp") and arr[k].ndim == 2]:
        p, margin = omas.monotonic_alignment_search(arr[k])
        print(f"  oracle MAS {k}: equal={np.array_equal(p, arr[k.replace('.logp', '.path')])} margin={margin:.2e}")


def gen_loss_tables(L):
    """Docstring tables of the reference as KATs + beta-binomial prior from scipy."""
    arr = {}
    ga = L.GuidedAttentionLoss
    arr["ga.mask_5_5"] = to_np(ga._make_guided_attention_mask(torch.tensor(5), torch.tensor(5), 0.4))
    arr["ga.mask_3_6"] = to_np(ga._make_guided_attention_mask(torch.tensor(3), torch.tensor(6), 0.4))
    fs = L.ForwardSumLoss()
    arr["fs.prior_7_4"] = to_np(fs._generate_prior(torch.tensor([4, 3]), torch.tensor([7, 5])))
    save("loss_tables", {"__model__": "losses"}, arr)
    from oracle import models as OM
    pr = OM.betabinom_logprior(7, 4)
    print(f"  betabinom prior err {np.abs(pr - arr['fs.prior_7_4'][0, :7, :4]).max():.2e}")


VTN_TINY = dict(idim=80, odim=80, dprenet_layers=2, dprenet_units=32, adim=32, aheads=2, elayers=2, eunits=64, dlayers=2,
                dunits=64, postnet_layers=5, postnet_filts=5, postnet_chans=32, decoder_reduction_factor=4,
                encoder_normalize_before=True, decoder_normalize_before=False)
VTN_CONF_TINY = dict(VTN_TINY, encoder_type="conformer", conformer_enc_kernel_size=7)
TTS_TINY = dict(idim=30, odim=80, dprenet_layers=2, dprenet_units=32, adim=32, aheads=2, elayers=2, eunits=64, dlayers=2,
                dunits=64, postnet_layers=5, postnet_filts=5, postnet_chans=32, decoder_reduction_factor=2)
AAS_TINY = dict(idim=80, odim=80, adim=32, aheads=2, elayers=2, eunits=64, dlayers=2, dunits=64, positionwise_layer_type="linear",
                positionwise_conv_kernel_size=1, duration_predictor_use_encoder_outputs=False, duration_predictor_input_dim=80,
                duration_predictor_layers=2, duration_predictor_chans=32, duration_predictor_kernel_size=3, postnet_layers=5,
                postnet_filts=5, postnet_chans=32, use_masking=True, encoder_normalize_before=True, decoder_normalize_before=True,
                encoder_reduction_factor=1, post_encoder_reduction_factor=4, decoder_reduction_factor=1, encoder_type="conformer",
                decoder_type="conformer", duration_predictor_type="stochastic", encoder_input_layer="linear",
                conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
                use_macaron_style_in_conformer=True, use_cnn_in_conformer=True, conformer_enc_kernel_size=7,
                conformer_dec_kernel_size=7, init_type="xavier_uniform")
FS2_TINY = dict(idim=80, odim=80, adim=32, aheads=2, elayers=2, eunits=64, dlayers=2, dunits=64, positionwise_layer_type="linear",
                positionwise_conv_kernel_size=1, duration_predictor_use_encoder_outputs=False, duration_predictor_input_dim=80,
                duration_predictor_layers=2, duration_predictor_chans=32, duration_predictor_kernel_size=3, postnet_layers=5,
                postnet_filts=5, postnet_chans=32, use_masking=True, encoder_normalize_before=True, decoder_normalize_before=True,
                encoder_reduction_factor=1, decoder_reduction_factor=1, encoder_type="conformer", decoder_type="conformer",
                encoder_input_layer="conv2d", conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
                use_macaron_style_in_conformer=True, use_cnn_in_conformer=True, conformer_enc_kernel_size=7,
                conformer_dec_kernel_size=7, init_type="xavier_uniform", teacher_model_decoder_reduction_factor=1)
AAS_DET_TINY = dict(AAS_TINY, duration_predictor_type="deterministic", duration_predictor_use_encoder_outputs=True,
                    post_encoder_reduction_factor=1, positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3)


def main():
    """python tools/gen_golden.py [fixture-name ...]  (no names: regenerate everything)"""
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    M, L, A = import_reference()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    want = lambda n: not only or n in only
    if want("mas_kats"):
        gen_mas_kats(A)
    if want("loss_tables"):
        gen_loss_tables(L)
    if want("vtn_tiny_train"):
        gen_vtn(M, L, "vtn_tiny_train", VTN_TINY, B=3, Ti=60, To=48, seed=100)
    if want("vtn_tiny_eval"):
        gen_vtn(M, L, "vtn_tiny_eval", VTN_TINY, B=2, Ti=41, To=37, seed=101, train=False)
    if want("vtn_conformer_tiny_train"):
        gen_vtn(M, L, "vtn_conformer_tiny_train", VTN_CONF_TINY, B=2, Ti=52, To=40, seed=102)
    if want("vtn_tiny_inference"):
        gen_vtn_inference(M, "vtn_tiny_inference", VTN_TINY, T=44, seed=103)
    if want("tts_tiny_train"):
        gen_tts(M, L, "tts_tiny_train", TTS_TINY, B=3, Ti=14, To=50, seed=104)
    if want("aasvc_tiny_train"):
        gen_aasvc(M, L, A, "aasvc_tiny_train", AAS_TINY, B=3, Ti=64, To=72, seed=105)
    if want("aasvc_det_tiny_train"):
        gen_aasvc(M, L, A, "aasvc_det_tiny_train", AAS_DET_TINY, B=2, Ti=48, To=60, seed=106)
    if want("vtn_preln_inference_stop"):       # pre-LN decoder, generation ended by the stop threshold, minlen > 0
        gen_vtn_inference(M, "vtn_preln_inference_stop", dict(VTN_TINY, decoder_normalize_before=True), T=68, seed=107,
                          maxlenratio=3.0, fire=True, minlenratio=1.6)
    if want("tts_tiny_inference"):
        gen_vtn_inference(M, "tts_tiny_inference", TTS_TINY, T=11, seed=108, maxlenratio=3.0, tts=True)
    if want("aasvc_tiny_inference"):           # decode path: no target
        gen_aasvc_inference(M, "aasvc_tiny_inference", AAS_TINY, T=52, seed=109)
    if want("aasvc_tiny_inference_gt"):        # debug path: with a target (alignment + durations returned as well)
        gen_aasvc_inference(M, "aasvc_tiny_inference_gt", AAS_TINY, T=64, seed=110, To=40)
    if want("aasvc_det_tiny_inference"):
        gen_aasvc_inference(M, "aasvc_det_tiny_inference", AAS_DET_TINY, T=37, seed=111)
    if want("fs2vc_tiny_train"):
        gen_fs2vc(M, L, "fs2vc_tiny_train", FS2_TINY, B=3, Ti=76, seed=112)
    for n in ("fw_enc384", "fw_dec384", "fw_conf384", "fw_conf1536"):
        if want(n):
            gen_fullwidth(n)


if __name__ == "__main__":
    main()
