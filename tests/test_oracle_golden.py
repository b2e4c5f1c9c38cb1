"""CPU: pin the oracle (oracle/*.py, oracle/mas.c) against golden vectors produced by the imported
reference (tools/gen_golden.py).  No GPU, no /root/reference needed at run time."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mas as omas
from oracle import models as OM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = json.loads(bytes(z["__cfg__"]).decode())
    return cfg, z


def sd_of(z):
    return {k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("sd.")}


def model_cfg(cfg):
    return {k: v for k, v in cfg.items() if not k.startswith("__")}


def close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(np.asarray(b)).double()
    same_inf = torch.isinf(a) & torch.isinf(b) & (a == b)
    err = torch.where(same_inf, torch.zeros_like(a), (a - b).abs()).max().item()
    assert err <= tol, f"max err {err:.3e} > {tol}"


def test_mas_known_answers():
    _, z = load("mas_kats")
    for k in [k for k in z.files if k.endswith(".logp") and z[k].ndim == 2]:
        path, margin = omas.monotonic_alignment_search(z[k])
        assert np.array_equal(path, z[k.replace(".logp", ".path")]), k
    ds, bl, _, _ = omas.viterbi_decode(z["kat4.logp"], z["kat4.text_lens"], z["kat4.feat_lens"])
    assert np.array_equal(ds, z["kat4.ds"])
    assert abs(bl - float(z["kat4.bin_loss"])) < 1e-6
    # survey KATs, literal expectations
    assert z["kat1.path"].tolist() == [0, 0, 1, 1, 2, 2]
    assert z["kat2.path"].tolist() == [0, 0, 0, 0, 1, 2]
    assert z["kat3.path"].tolist() == [2, 3, 4]


def test_mas_c_restatement_matches_numpy():
    cmas = pytest.importorskip("oracle.cmas")
    if not cmas.available():
        pytest.skip("oracle/libmas_oracle.so not built")
    _, z = load("mas_kats")
    for k in [k for k in z.files if k.endswith(".logp") and z[k].ndim == 2]:
        assert np.array_equal(cmas.mas(z[k]), z[k.replace(".logp", ".path")]), k


def test_loss_tables():
    _, z = load("loss_tables")
    att = torch.ones(1, 1, 5, 5)
    il, ol = torch.tensor([5]), torch.tensor([5])
    # mean of W over the valid region equals the oracle loss on an all-ones attention map
    close(OM.guided_attention_loss(att, il, ol), z["ga.mask_5_5"].mean(), 1e-6)
    att = torch.ones(1, 1, 6, 3)
    close(OM.guided_attention_loss(att, torch.tensor([3]), torch.tensor([6])), z["ga.mask_3_6"].mean(), 1e-6)
    # docstring table of the reference (losses/guided_attention_loss.py:71-90)
    assert abs(float(z["ga.mask_5_5"][0, 1]) - 0.1175) < 1e-4 and abs(float(z["ga.mask_3_6"][5, 0]) - 0.8858) < 1e-4
    close(OM.betabinom_logprior(7, 4), z["fs.prior_7_4"][0, :7, :4], 1e-6)
    close(OM.betabinom_logprior(5, 3), z["fs.prior_7_4"][1, :5, :3], 1e-6)


@pytest.mark.parametrize("name", ["vtn_tiny_train", "vtn_tiny_eval", "vtn_conformer_tiny_train"])
def test_vtn_forward_and_grads(name):
    cfg, z = load(name)
    sd = sd_of(z)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    t = lambda k: torch.from_numpy(z[k])
    o = OM.vtn_forward(sd, model_cfg(cfg), t("in.xs"), t("in.ilens"), t("in.ys"), t("in.labels"), t("in.olens"),
                       training=cfg["__train__"])
    close(o[0], z["out.after"]); close(o[1], z["out.before"]); close(o[2], z["out.logits"])
    close(o[3], z["out.ys"], 0); close(o[4], z["out.labels"], 0)
    assert torch.equal(o[5], t("out.olens")) and torch.equal(o[6][1], t("out.ilens_ds")) and torch.equal(o[6][2], t("out.olens_in"))
    for i, a in enumerate(o[6][0]):
        close(a, z[f"out.att_ws.{i}"], 1e-6)
    l1, bce = OM.seq2seq_loss(o[0], o[1], o[2], o[3], o[4], o[5])
    close(l1, z["loss.l1"], 1e-6); close(bce, z["loss.bce"], 1e-6)
    close(OM.guided_attention_loss(o[6][0][0], o[6][1], o[6][2]), z["loss.guided_attn"], 1e-6)
    (l1 + bce).backward()
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(params[k[5:]].grad, z[k], 5e-5)
    for k in [k for k in z.files if k.startswith("sd_after.")]:
        close(sd[k[9:]].detach(), z[k], 1e-5)


@pytest.mark.parametrize("name", ["vtn_tiny_inference", "vtn_preln_inference_stop", "tts_tiny_inference"])
def test_ar_inference(name):
    """Generation loop: run to maxlen / stop through the threshold with minlen in force (pre-LN decoder) / TTS."""
    cfg, z = load(name)
    with torch.no_grad():
        outs, probs, att = OM.vtn_inference(sd_of(z), model_cfg(cfg), torch.from_numpy(z["in.x"]),
                                            tts=cfg["__model__"] == "TransformerTTS", **cfg["__inference__"])
    assert tuple(outs.shape) == z["out.outs"].shape      # same stop step as the reference
    close(outs, z["out.outs"], 1e-5); close(probs, z["out.probs"], 1e-6); close(att, z["out.att_ws"], 1e-6)


def test_tts_forward_and_grads():
    cfg, z = load("tts_tiny_train")
    sd = sd_of(z)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    t = lambda k: torch.from_numpy(z[k])
    o = OM.tts_forward(sd, model_cfg(cfg), t("in.xs"), t("in.ilens"), t("in.ys"), t("in.labels"), t("in.olens"))
    close(o[0], z["out.after"]); close(o[1], z["out.before"]); close(o[2], z["out.logits"])
    l1, bce = OM.seq2seq_loss(o[0], o[1], o[2], o[3], o[4], o[5])
    close(l1, z["loss.l1"], 1e-6); close(bce, z["loss.bce"], 1e-6)
    (l1 + bce).backward()
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(params[k[5:]].grad, z[k], 5e-5)


@pytest.mark.parametrize("name", ["aasvc_tiny_train", "aasvc_det_tiny_train"])
def test_aasvc_forward_and_grads(name):
    cfg, z = load(name)
    sd = sd_of(z)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    t = lambda k: torch.from_numpy(z[k])
    noise = t("in.sdp_noise") if "in.sdp_noise" in z.files else None
    r = OM.aasvc_forward(sd, model_cfg(cfg), t("in.xs"), t("in.ilens"), t("in.ys"), t("in.olens"), dp_inputs=t("in.xs"), noise=noise)
    close(r["log_p_attn"], z["out.log_p_attn"]); close(r["after_outs"], z["out.after"]); close(r["before_outs"], z["out.before"])
    assert torch.equal(r["ds"], t("out.ds")), "durations must be bit-exact"
    for b, pth in enumerate(r["mas_paths"]):
        assert np.array_equal(pth, z[f"out.mas_path.{b}"]), "alignment indices must be bit-exact"
    close(r["bin_loss"], z["out.bin_loss"], 1e-6)
    l1 = OM.l1_loss(r["after_outs"], r["before_outs"], r["ys"], r["olens"])
    fs = OM.forward_sum_loss(r["log_p_attn"], r["ilens"], r["olens_reduced"])
    close(l1, z["loss.l1"], 1e-6); close(fs, z["loss.forward_sum"], 1e-5)
    total = l1 + cfg["__lambda_align__"] * (fs + r["bin_loss"])
    if "dur_nll" in r:
        close(r["dur_nll"], z["out.dur_nll"], 1e-4)
        total = total + r["dur_nll"].sum()
    else:
        close(r["d_outs"], z["out.d_outs"])
    close(total, z["loss.total"], 1e-4)
    total.backward()
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(params[k[5:]].grad, z[k], 2e-4)


@pytest.mark.parametrize("name", ["aasvc_tiny_inference", "aasvc_tiny_inference_gt", "aasvc_det_tiny_inference"])
def test_aasvc_inference(name):
    """AASVC.inference (models/aas_vc.py:531-603): decode path without a target, debug path with one."""
    cfg, z = load(name)
    t = lambda k: torch.from_numpy(z[k])
    x = t("in.x")
    y = t("in.y") if "in.y" in z.files else None
    noise = t("in.sdp_noise") if "in.sdp_noise" in z.files else None
    with torch.no_grad():
        r = OM.aasvc_forward(sd_of(z), model_cfg(cfg), x[None], torch.tensor([x.shape[0]]), None if y is None else y[None],
                             None if y is None else torch.tensor([y.shape[0]]), dp_inputs=x[None], noise=noise,
                             training=False, inference=True)
    assert torch.equal(r["d_outs"][0].float(), t("out.d_outs").float()), "predicted durations must be identical"
    assert tuple(r["after_outs"][0].shape) == z["out.outs"].shape
    close(r["after_outs"][0], z["out.outs"], 1e-5)
    if y is not None:
        assert torch.equal(r["ds"][0], t("out.ds"))
        close(r["log_p_attn"][0], z["out.log_p_attn"], 1e-5)
        assert int(r["ilens"][0]) == int(z["out.ilens"])


@pytest.mark.parametrize("name", ["fw_enc384", "fw_dec384", "fw_conf384", "fw_conf1536"])
def test_fullwidth_layers(name):
    """Full-width single layers (tests/fullwidth.py): the seeded weights reproduce the generator's checksums and the
    oracle's layer functions reproduce the reference layer's output and gradients at d=384 / d=1536."""
    import fullwidth as FW
    from oracle import nets as N
    cfg, z = load(name)
    c = FW.CASES[name]
    assert {k: v for k, v in cfg.items() if not k.startswith("__")} == c
    names = [k[6:] for k in z.files if k.startswith("chk.w.")]
    shapes = FW.layer_param_shapes(c)
    assert [n for n, _ in shapes] == names
    state = FW.seeded_state(shapes, c["seed"])
    for k in names:
        assert FW.checksum(state[k]) == int(z["chk.w." + k]), k
    x, mem, dy = FW.inputs(c)
    assert FW.checksum(x) == int(z["chk.x"]) and FW.checksum(dy) == int(z["chk.dy"])
    sd = {k: v.requires_grad_(True) for k, v in state.items()}
    for k in [k for k in z.files if k.startswith("buf.")]:
        sd[k[4:]] = torch.from_numpy(z[k]).clone()
        if "running_mean" in k:
            sd[k[4:]].zero_()
        elif "running_var" in k:
            sd[k[4:]].fill_(1.0)
        else:
            sd[k[4:]].zero_()
    out = FW.oracle_layer(c, sd, x.requires_grad_(True), mem.requires_grad_(True) if mem is not None else None)
    close(out, z["out"], 2e-5)
    (out * dy).sum().backward()
    close(x.grad.reshape(-1)[::3], z["dx"], 2e-5)
    if mem is not None:
        close(mem.grad.reshape(-1)[::3], z["dmem"], 2e-5)
    for k in names:
        close(sd[k].grad.reshape(-1)[::FW.grad_stride(sd[k].numel())], z["grad." + k], 1e-4)


def test_fastspeech_vc_forward_grads_inference_and_duration_calculator():
    """FastSpeechVC (models/fastspeech_vc.py:244-466): training forward, L1 + duration loss, gradients, the inference path;
    DurationCalculator (utils/duration_calculator.py:13-65) known answers for the 4-D and the 2-D case."""
    cfg, z = load("fs2vc_tiny_train")
    mc = model_cfg(cfg)
    sd = sd_of(z)
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    t = lambda k: torch.from_numpy(z[k])
    o = OM.fastspeech_vc_forward(sd, mc, t("in.xs"), t("in.ilens"), t("in.ys"), t("in.olens"), t("in.ds"), dp_inputs=t("in.xs"))
    close(o[0], z["out.before"]); close(o[1], z["out.after"]); close(o[2], z["out.d_outs"])
    assert torch.equal(o[3], t("out.ilens")) and torch.equal(o[4], t("out.olens"))
    l1 = OM.l1_loss(o[1], o[0], o[5], o[4])
    dl = OM.duration_predictor_loss(o[2], t("in.ds"), o[3])
    close(l1, z["loss.l1"], 1e-6); close(dl, z["loss.duration"], 1e-6)
    (l1 + dl).backward()
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(params[k[5:]].grad, z[k], 1e-4)
    for k in [k for k in z.files if k.startswith("sd_after.")]:
        close(sd[k[9:]].detach(), z[k], 1e-5)
    sd_inf = {k: v.detach() for k, v in sd.items()}
    x = t("inf.x")
    with torch.no_grad():
        oi = OM.fastspeech_vc_forward(sd_inf, mc, x[None], torch.tensor([x.shape[0]]), dp_inputs=x[None], training=False, inference=True)
    assert torch.equal(oi[2][0], t("inf.d_outs"))
    close(oi[1][0], z["inf.outs"], 1e-5)
    d4, f4 = OM.duration_calculator(t("dc.att4"))
    d2, f2 = OM.duration_calculator(t("dc.att2"))
    assert torch.equal(d4, t("dc.dur4")) and torch.equal(d2, t("dc.dur2"))
    close(f4, z["dc.focus4"], 1e-7); close(f2, z["dc.focus2"], 1e-7)
    assert int(d4.sum()) == 37 and int(d2.sum()) == 29          # every output frame is counted exactly once


def test_logmel_analytic_known_answers():
    """log-mel front-end (reference bin/preprocess.py:30-92 -> librosa, absent here: parity stays formally unpinned).  The
    restatement is pinned against closed forms instead: (1) a bin-centred cosine of amplitude A has, under a periodic Hann
    window of length N with no normalisation, |X[k]| = A*N/4, |X[k+-1]| = A*N/8 and nothing else -- which fixes window type,
    window periodicity, STFT scaling and magnitude-vs-power; (2) a constant signal survives the centre/reflect padding
    unchanged (frame 0 is centred on sample 0): |X[0]| = c*N/2, |X[1]| = c*N/4; (3) a unit impulse gives a flat magnitude
    spectrum equal to the window value at its position, so mel_m = w * sum_k fb[m, k], and the Slaney area normalisation makes
    that sum ~ n_fft / sr for every filter wide enough to be sampled; (4) frames = 1 + N // hop; (5) the float32 chain agrees
    with the float64 chain to 1e-4 in log10 units on speech-like noise."""
    from oracle import logmel as LM
    sr, N, hop = 16000, 1024, 256
    kw = dict(fft_size=N, hop_size=hop, num_mels=80, fmin=80, fmax=7600)
    fb = LM.mel_filterbank64(sr, N, 80, 80, 7600)
    n = hop * 40
    # (1) bin-centred cosine
    k0, A = 100, 0.37
    x = A * np.cos(2 * np.pi * k0 * np.arange(n) / N + 0.3)
    mag = np.zeros(N // 2 + 1)
    mag[k0], mag[k0 - 1], mag[k0 + 1] = A * N / 4, A * N / 8, A * N / 8
    want = np.log10(np.maximum(1e-10, fb @ mag))
    got = LM.logmelfilterbank(x, sr, dtype=np.float64, **kw)
    assert got.shape == (1 + n // hop, 80)                                    # (4)
    inner = got[8:-8]                                                          # frames untouched by the reflect padding
    hit = want > -5                                                            # filters that see the tone
    assert hit.sum() >= 2 and np.abs(inner[:, hit] - want[hit]).max() < 1e-9
    assert (inner[:, ~hit] < -8).all()                                         # everything else is rounding noise (eps floor at -10)
    got32 = LM.logmelfilterbank(x.astype(np.float32), sr, **kw)
    assert np.abs(got32[8:-8][:, hit] - want[hit]).max() < 1e-4
    # (2) constant signal, first frame (centre padding by reflection)
    c = 0.25
    mag = np.zeros(N // 2 + 1)
    mag[0], mag[1] = c * N / 2, c * N / 4
    want = np.log10(np.maximum(1e-10, fb @ mag))
    got = LM.logmelfilterbank(np.full(n, c), sr, dtype=np.float64, **kw)
    assert np.abs(got[0] - want).max() < 1e-9 and np.abs(got[20] - want).max() < 1e-9
    # (3) impulse: flat spectrum scaled by the window value at its position in the frame
    x = np.zeros(n)
    p = hop * 20 + 77
    x[p] = 1.0
    got = LM.logmelfilterbank(x, sr, dtype=np.float64, **kw)
    t = 20                                                                     # frame t covers samples [t*hop - N/2, t*hop + N/2)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * (p - (t * hop - N // 2)) / N)
    assert np.abs(got[t] - np.log10(w * fb.sum(1))).max() < 1e-9
    wide = (fb > 0).sum(1) >= 8
    assert wide.sum() > 30 and np.abs(fb.sum(1)[wide] / (N / sr) - 1).max() < 0.02
    # (5) fp32 adequacy on noise with a speech-like spectral tilt
    rng = np.random.default_rng(0)
    x = np.cumsum(rng.standard_normal(n)) * 0.01
    x = (x - x.mean()) / (np.abs(x).max() + 1e-9) * 0.5
    d = np.abs(LM.logmelfilterbank(x.astype(np.float32), sr, **kw) - LM.logmelfilterbank(x.astype(np.float32).astype(np.float64), sr, dtype=np.float64, **kw))
    assert d.max() < 1e-4, d.max()


def test_logmel_against_independent_stft_implementations():
    """The STFT of oracle/logmel.py (= librosa.stft(center=True, pad_mode="reflect", window="hann") restated,
    bin/preprocess.py:63-70) against two INDEPENDENT third-party implementations present in the image -- torch.stft and
    scipy.signal.stft -- on random audio whose length is / is not a multiple of the hop, and the whole log-mel chain through
    them.  (librosa itself is absent: this pins the restatement against the same algorithm as implemented by others, which is as
    close as this container gets; a librosa-specific deviation, if there is one, stays invisible here.)"""
    import scipy.signal as ss
    import torch
    from oracle import logmel as LM
    sr, N, hop = 16000, 1024, 256
    fb = LM.mel_filterbank64(sr, N, 80, 80, 7600)
    rng = np.random.default_rng(7)
    for n in (hop * 31, hop * 31 + 1, hop * 40 + 255, 16000 * 2 + 77, N // 2 + 5):
        x = (rng.standard_normal(n) * 0.1).astype(np.float64)
        # the restatement's own magnitude spectrogram (float64)
        xp = np.pad(x, N // 2, mode="reflect")
        frames = 1 + n // hop
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
        idx = np.arange(N)[None, :] + hop * np.arange(frames)[:, None]
        mine = np.abs(np.fft.rfft(xp[idx] * win, axis=1))
        # torch.stft: centre padding, reflect, periodic Hann, no normalisation
        t = torch.stft(torch.from_numpy(x), n_fft=N, hop_length=hop, win_length=N, window=torch.hann_window(N, periodic=True, dtype=torch.float64),
                       center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().T.numpy()
        assert t.shape == mine.shape == (frames, N // 2 + 1), (t.shape, mine.shape, n)
        assert np.abs(t - mine).max() < 1e-9 * max(1.0, mine.max()), (n, np.abs(t - mine).max())
        # scipy.signal.stft on the reflect-padded signal, frames at multiples of the hop, `spectrum` scaling undone
        w = ss.get_window("hann", N, fftbins=True)
        assert np.abs(w - win).max() < 1e-15
        _, _, Z = ss.stft(xp, fs=sr, window=w, nperseg=N, noverlap=N - hop, nfft=N, boundary=None, padded=False, return_onesided=True)
        sc = np.abs(Z).T * w.sum()
        assert sc.shape == mine.shape and np.abs(sc - mine).max() < 1e-9 * max(1.0, mine.max()), (n, sc.shape)
        # the whole chain through the third-party STFT
        want = np.log10(np.maximum(1e-10, t @ fb.T))
        got = LM.logmelfilterbank(x, sr, fft_size=N, hop_size=hop, num_mels=80, fmin=80, fmax=7600, dtype=np.float64)
        assert np.abs(got - want).max() < 1e-9
        got32 = LM.logmelfilterbank(x.astype(np.float32), sr, fft_size=N, hop_size=hop, num_mels=80, fmin=80, fmax=7600)
        assert np.abs(got32 - want).max() < 2e-4


def test_logmel_against_the_librosa_port_of_transformers():
    """A third implementation of the two librosa calls the reference makes (bin/preprocess.py:63-82): `transformers.audio_utils`
    (`mel_filter_bank(norm="slaney", mel_scale="slaney")`, `spectrogram(center=True, pad_mode="reflect", power=1.0, log_mel="log10")`),
    which its authors adapted from librosa and check against it.  oracle/logmel.py agrees with it to rounding in float64 at the
    reference's recipe settings (16 kHz / 24 kHz, fmin 80, fmax 7600) -- not a pin in the sense of the reference's own vectors (the
    header of oracle/logmel.py keeps saying "parity unpinned"), but the restatement and an independent port of librosa meet."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    from oracle import logmel as LM
    rng = np.random.default_rng(11)
    for sr, nfft, hop, nm, fmin, fmax in ((16000, 1024, 256, 80, 80, 7600), (24000, 2048, 300, 80, 80, 7600), (22050, 1024, 256, 80, 0, None)):
        fb_t = audio_utils.mel_filter_bank(nfft // 2 + 1, nm, float(fmin), float(fmax if fmax else sr / 2), sr, norm="slaney", mel_scale="slaney")
        fb_o = LM.mel_filterbank64(sr, nfft, nm, fmin, fmax if fmax else sr / 2)
        assert np.abs(fb_o - fb_t.T).max() < 1e-14
        n = sr + 123
        x = rng.standard_normal(n) * 0.1 + 0.3 * np.sin(2 * np.pi * 440.0 * np.arange(n) / sr)
        ours = LM.logmelfilterbank(x, sr, nfft, hop, nm, fmin, fmax, dtype=np.float64)
        win = audio_utils.window_function(nfft, "hann", periodic=True)
        theirs = audio_utils.spectrogram(x, win, nfft, hop, nfft, power=1.0, center=True, pad_mode="reflect", mel_filters=fb_t,
                                         mel_floor=1e-10, log_mel="log10", dtype=np.float64).T
        assert ours.shape == theirs.shape, (ours.shape, theirs.shape)
        assert np.abs(ours - theirs).max() < 1e-6, np.abs(ours - theirs).max()


def test_mel_basis_against_the_published_slaney_formula():
    """librosa.filters.mel(htk=False, norm="slaney") (bin/preprocess.py:76-82) = the mel scale of Slaney's Auditory Toolbox --
    linear below 1 kHz at 200/3 Hz per mel, above it 27 steps per factor 6.4 -- and triangles of unit AREA.  Checked: the
    published constants; a second, differently written evaluation (np.interp over the three corner frequencies, float64);
    the product's own basis (seq2seq_vc_amd/frontend.py: librosa's ramp formulation); support, peak and area of every filter."""
    from oracle import logmel as LM
    from seq2seq_vc_amd.frontend import mel_basis
    assert abs(float(LM._hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(LM._hz_to_mel(6400.0)) - 42.0) < 1e-9
    assert abs(float(LM._hz_to_mel(200.0)) - 3.0) < 1e-12 and abs(float(LM._mel_to_hz(42.0)) - 6400.0) < 1e-6
    assert abs(float(LM._mel_to_hz(LM._hz_to_mel(3333.0))) - 3333.0) < 1e-8
    for sr, N, M, fmin, fmax in ((16000, 1024, 80, 80, 7600), (24000, 2048, 80, 0, 12000), (22050, 1024, 40, 0, 11025)):
        freqs = np.arange(N // 2 + 1) * (sr / N)
        pts = LM._mel_to_hz(np.linspace(LM._hz_to_mel(fmin), LM._hz_to_mel(fmax), M + 2))
        second = np.stack([np.interp(freqs, pts[i:i + 3], [0.0, 1.0, 0.0], left=0.0, right=0.0) * 2.0 / (pts[i + 2] - pts[i]) for i in range(M)])
        a = LM.mel_filterbank64(sr, N, M, fmin, fmax)
        assert np.abs(a - second).max() < 1e-12
        assert np.abs(LM.mel_filterbank(sr, N, M, fmin, fmax) - second).max() < 1e-9          # the float32 cast only
        assert np.abs(mel_basis(sr, N, M, fmin, fmax).astype(np.float64) - second).max() < 1e-9
        for i in range(M):
            nz = np.nonzero(a[i])[0]
            assert (a[i] >= 0).all() and (freqs[nz] > pts[i]).all() and (freqs[nz] < pts[i + 2]).all()
            if len(nz) >= 8:                                                                       # wide enough to be sampled
                assert abs(a[i].sum() * (sr / N) - 1.0) < 0.03                                     # unit area (Slaney normalisation)
                assert abs(freqs[np.argmax(a[i])] - pts[i + 1]) <= sr / N                          # peak at the centre frequency


def test_every_recipe_config_of_the_reference_builds_the_same_state_dict():
    """"run.sh recipes are drop-in".  tests/golden/recipe_configs.json (tools/gen_recipe_configs.py) holds, for each of
    the reference's 13 recipe YAMLs that name a model of the path, the file's model_params and the state_dict keys / shapes / dtypes
    of the REFERENCE's model built from them.  The product's class of the same name, built from the same params, must give the same
    keys in the same order with the same shapes and dtypes (checkpoints interchange by key; `init-mods` prefixes resolve), and the
    recipe's trainer / collater / criterion names must resolve in seq2seq_vc_amd."""
    import json
    import seq2seq_vc_amd.collaters as C
    import seq2seq_vc_amd.losses as L
    import seq2seq_vc_amd.models as M
    import seq2seq_vc_amd.trainers as T
    with open(os.path.join(GOLD, "recipe_configs.json")) as f:
        recipes = json.load(f)
    assert len(recipes) == 13
    seen = set()
    for r in recipes:
        torch.manual_seed(0)
        model = getattr(M, r["model_type"])(**r["model_params"], **r["injected_params"])
        sd = model.state_dict()
        got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        assert [g[0] for g in got] == [e[0] for e in r["state_dict"]], f"{r['recipe']}: state_dict keys / order differ"
        bad = [(g, e) for g, e in zip(got, r["state_dict"]) if g != e]
        assert not bad, f"{r['recipe']}: {bad[:3]}"
        assert sum(p.numel() for p in model.parameters() if p.requires_grad) == r["n_trainable"], r["recipe"]
        for prefix in (r["init_mods"] or []):                    # partial loading by module prefix (utils/model_io.py:12-111)
            assert any(k.startswith(prefix + ".") for k in sd), f"{r['recipe']}: init-mods prefix {prefix} matches nothing"
        if r["trainer_type"]:
            assert hasattr(T, r["trainer_type"]), f"{r['recipe']}: trainer {r['trainer_type']}"
        if r["collater_type"]:
            assert hasattr(C, r["collater_type"]), f"{r['recipe']}: collater {r['collater_type']}"
        for name, kw in (r["criterions"] or {}).items():
            getattr(L, name)(**(kw or {}))
        seen.add(r["model_type"])
    assert seen == {"VTN", "AASVC", "TransformerTTS", "FastSpeechVC"}
