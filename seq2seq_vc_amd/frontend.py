"""STFT -> log-mel feature front-end on the GPU (drop-in for `logmelfilterbank`, reference
bin/preprocess.py:30-92, and -- optionally fused -- the mean/variance normalisation of
bin/normalize.py:172-193).

    mel = logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600)

librosa semantics restated (librosa itself is an unpinned third-party dependency of the reference and is not
installed here, so this path is "parity unpinned": it is checked against the independent numpy restatement
in oracle/logmel.py, not against librosa): center=True with reflect padding of n_fft/2, periodic Hann window,
frames = 1 + N // hop, one-sided spectrum of n_fft/2+1 bins, magnitude, Slaney-scale area-normalised mel
basis, max(eps, .), log10.

GPU formulation: the (frames x n_fft) frame matrix is never built -- the padded signal IS the A operand of the
fp32 MFMA GEMM with leading dimension `hop` (overlapping rows); B is the windowed real-DFT basis
[cos | -sin] (2*(n_fft/2+1) x n_fft, fp32, cached on the device).  A second small GEMM applies the mel basis.
"""
import math

import numpy as np
import torch

from . import _lib
from .ops import kernels as K

_BASIS = {}


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=float)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=float)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """Slaney mel filterbank, area-normalised (librosa.filters.mel defaults) -> (n_mels, n_fft//2+1) float32."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def dft_basis(n_fft, win_length=None):
    """Windowed one-sided DFT basis rows [w*cos ; -w*sin] -> (2*(n_fft//2+1), n_fft) float32."""
    win_length = n_fft if win_length is None else win_length
    k = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * k / win_length)         # periodic Hann (fftbins=True)
    lp = (n_fft - win_length) // 2
    w = np.zeros(n_fft)
    w[lp:lp + win_length] = win
    n = np.arange(n_fft // 2 + 1)[:, None]
    ang = 2 * np.pi * n * np.arange(n_fft)[None, :] / n_fft
    return np.concatenate([np.cos(ang) * w, -np.sin(ang) * w], axis=0).astype(np.float32)


def _tables(device, sr, n_fft, win_length, n_mels, fmin, fmax):
    key = (str(device), sr, n_fft, win_length, n_mels, fmin, fmax)
    if key not in _BASIS:
        _BASIS[key] = (torch.from_numpy(dft_basis(n_fft, win_length)).to(device),
                       torch.from_numpy(mel_basis(sr, n_fft, n_mels, fmin, fmax)).to(device))
    return _BASIS[key]


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, log_base=10.0, mean=None, scale=None, impl=None):
    """audio: 1-D float tensor on the GPU (or numpy/CPU tensor, copied once) -> (frames, num_mels) fp32 tensor
    on the GPU.  `mean`/`scale` (num_mels,) fuse `(x - mean) / scale` into the log kernel.
    impl: None / "fft" = the one-launch FFT kernel (csrc/stft_fft.hip) where it applies (n_fft = 1024 with triangular mel filters: its
    radix-8 form), "fft_radix4" = the generic radix-4 kernel for any of its sizes, "gemm" = the DFT-as-GEMM path (5 launches)."""
    if window != "hann":
        raise NotImplementedError("only the hann window of the recipes is supported")
    if not isinstance(audio, torch.Tensor):
        audio = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    if not audio.is_cuda:
        audio = audio.cuda()
    audio = audio.float().contiguous()
    dev = audio.device
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    if log_base not in (None, 10.0, 2.0):
        raise ValueError(f"{log_base} is not supported.")
    wl_ = fft_size if win_length is None else win_length
    if impl != "gemm" and fft_size in (512, 1024, 2048) and wl_ <= fft_size and audio.numel() > 0:
        n_ = audio.numel()
        fr_ = 1 + n_ // hop_size
        mt = ist = None
        if mean is not None:
            mt = torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous()
            ist = (1.0 / torch.as_tensor(scale, dtype=torch.float32, device=dev)).contiguous()
        out = stft_logmel_fft_device(audio.view(1, n_), torch.tensor([n_], dtype=torch.int32, device=dev),
                                     torch.tensor([fr_], dtype=torch.int32, device=dev), fr_, sampling_rate, fft_size, hop_size, win_length,
                                     num_mels, fmin, fmax, eps, 1.0 if log_base is None else 1.0 / math.log(log_base), mt, ist,
                                     radix8=False if impl == "fft_radix4" else None)
        return out[0]
    basis, melb = _tables(dev, sampling_rate, fft_size, win_length, num_mels, fmin, fmax)
    n = audio.numel()
    pad = fft_size // 2
    frames = 1 + n // hop_size
    nb = fft_size // 2 + 1
    L = _lib.lib()
    st = K.stream()
    padded = torch.empty(n + 2 * pad, dtype=torch.float32, device=dev)
    _lib.check(L.s2svc_reflect_pad(n, pad, audio.data_ptr(), padded.data_ptr(), st), "reflect_pad")
    z = torch.empty((frames, 2 * nb), dtype=torch.float32, device=dev)
    K.gemm(K.operand(padded, hop_size), K.operand(basis, fft_size), frames, 2 * nb, fft_size, z, in_dtype=torch.float32)
    spc = torch.empty((frames, nb), dtype=torch.float32, device=dev)
    _lib.check(L.s2svc_magnitude(frames, nb, z.data_ptr(), spc.data_ptr(), st), "magnitude")
    mel = torch.empty((frames, num_mels), dtype=torch.float32, device=dev)
    K.gemm(K.operand(spc, nb), K.operand(melb, nb), frames, num_mels, nb, mel, in_dtype=torch.float32)
    inv_log = 1.0 if log_base is None else 1.0 / math.log(log_base)
    if log_base not in (None, 10.0, 2.0):
        raise ValueError(f"{log_base} is not supported.")
    out = torch.empty_like(mel)
    m_ptr = s_ptr = None
    if mean is not None:
        mean_t = torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous()
        inv_scale_t = (1.0 / torch.as_tensor(scale, dtype=torch.float32, device=dev)).contiguous()
        m_ptr, s_ptr = mean_t.data_ptr(), inv_scale_t.data_ptr()
    _lib.check(L.s2svc_log_clamp(mel.numel(), num_mels, mel.data_ptr(), eps, inv_log, m_ptr, s_ptr, out.data_ptr(), st), "log_clamp")
    return out


def _mel_ranges(melb_np):
    """[lo, hi) of the non-zero bins of every mel filter."""
    nz = melb_np > 0
    lo = np.where(nz.any(1), nz.argmax(1), 0)
    hi = np.where(nz.any(1), melb_np.shape[1] - nz[:, ::-1].argmax(1), 0)
    return lo.astype(np.int32), hi.astype(np.int32)


_RANGES = {}


_FFT_TABLES = {}


def _fft_tables(device, sr, n_fft, win_length, n_mels, fmin, fmax):
    """Tables of the FFT-in-LDS kernel (csrc/stft_fft.hip), built in float64 and rounded once: window (zero-padded to n_fft),
    twiddles of the n_fft/2-point FFT and of the real-FFT unpack step, the mel filters' non-zero weights back to back."""
    key = (str(device), sr, n_fft, win_length, n_mels, fmin, fmax)
    if key not in _FFT_TABLES:
        wl = n_fft if win_length is None else win_length
        win = np.zeros(n_fft)
        lp = (n_fft - wl) // 2
        win[lp:lp + wl] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(wl) / wl)
        h = n_fft // 2
        a = -2 * np.pi * np.arange(h) / h
        w_half = np.stack([np.cos(a), np.sin(a)], 1)
        a = -2 * np.pi * np.arange(h + 1) / n_fft
        w_full = np.stack([np.cos(a), np.sin(a)], 1)
        melb = mel_basis(sr, n_fft, n_mels, fmin, fmax)
        lo, hi = _mel_ranges(melb)
        off = np.concatenate([[0], np.cumsum(hi - lo)]).astype(np.int32)
        maxw = int((hi - lo).max()) + 1                   # (+1: the kernel's loop advances two bins at a time)
        melw = np.concatenate([melb[m, lo[m]:hi[m]] for m in range(n_mels)] + [np.zeros(maxw + 2, np.float32)]).astype(np.float32)
        t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x.astype(dt))).to(device)
        melw_n = int(len(melw))
        packed = np.concatenate([w_half.reshape(-1), w_full.reshape(-1), win, melw, np.zeros(melw_n % 2)]).astype(np.float32)
        packed = np.concatenate([packed, np.zeros((-len(packed)) % 4, np.float32)])           # whole 16-byte vectors
        _FFT_TABLES[key] = (t(packed, np.float32), t(lo, np.int32), t(hi, np.int32), t(off[:-1], np.int32), melw_n, maxw)
    return _FFT_TABLES[key]


_FFT8_TABLES = {}


def _fft8_tables(device, sr, n_fft, n_mels, fmin, fmax):
    """Segment form of the mel basis for the radix-8 kernel (s2svc_stft_logmel_fft8), or None if the basis does not have the
    structure it needs: every bin has at most two non-zero weights, in neighbouring filters (true for librosa's triangular
    filters).  Segment s (0 .. n_mels) = the bins between the peaks of filters s - 1 and s: rising side of filter s (weight
    wud[k][0]), falling side of filter s - 1 (wud[k][1]); mel[m] = sum over segment m of wud[:, 0] |X| + sum over segment m + 1 of
    wud[:, 1] |X|.  -> (seg_lo, seg_len (n_mels + 1), wud (bins, 2))."""
    key = (str(device), sr, n_fft, n_mels, fmin, fmax)
    if key not in _FFT8_TABLES:
        melb = np.asarray(mel_basis(sr, n_fft, n_mels, fmin, fmax), dtype=np.float32)
        nb = melb.shape[1]
        peak = melb.argmax(axis=1)
        seg = np.full(nb, -1, np.int64)
        wud = np.zeros((nb, 2), np.float32)
        ok = n_fft == 1024 and n_mels <= 128
        for k in range(nb):
            nz = np.nonzero(melb[:, k])[0]
            if len(nz) == 0 or not ok:
                continue
            if len(nz) > 2 or (len(nz) == 2 and nz[1] != nz[0] + 1):
                ok = False
            elif len(nz) == 2:
                seg[k], wud[k, 0], wud[k, 1] = nz[1], melb[nz[1], k], melb[nz[0], k]
            elif k <= peak[nz[0]]:                     # one filter only, on its rising side (or at its peak)
                seg[k], wud[k, 0] = nz[0], melb[nz[0], k]
            else:                                      # ... on its falling side: the segment above
                seg[k], wud[k, 1] = nz[0] + 1, melb[nz[0], k]
        seg_lo, seg_len = np.zeros(n_mels + 1, np.int32), np.zeros(n_mels + 1, np.int32)
        if ok:
            for m in range(n_mels + 1):
                ks = np.nonzero(seg == m)[0]
                if len(ks) == 0:
                    continue
                if ks[-1] - ks[0] + 1 != len(ks):
                    ok = False
                    break
                seg_lo[m], seg_len[m] = ks[0], len(ks)
        if ok:      # the identity the kernel relies on, checked on the tables themselves
            rec = np.zeros_like(melb)
            for m in range(n_mels):
                sl = slice(seg_lo[m], seg_lo[m] + seg_len[m])
                rec[m, sl] += wud[sl, 0]
                sl = slice(seg_lo[m + 1], seg_lo[m + 1] + seg_len[m + 1])
                rec[m, sl] += wud[sl, 1]
            ok = bool(np.array_equal(rec, melb))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        _FFT8_TABLES[key] = (t(seg_lo), t(seg_len), t(wud)) if ok else None
    return _FFT8_TABLES[key]


def stft_logmel_fft_device(x, nlen_d, frames_d, Tmax, sampling_rate, fft_size, hop_size, win_length, num_mels, fmin, fmax, eps, inv_log,
                           mean_t=None, inv_scale_t=None, out=None, radix8=None):
    """The ONE launch of the FFT front-end on device-resident arguments (x (B, Nmax) fp32, nlen_d / frames_d (B) int32):
    capturable, no host work beyond the launch.  -> (B, Tmax, num_mels) fp32."""
    dev = x.device
    B, Nmax = x.shape
    tables, lo, hi, off, melw_n, maxw = _fft_tables(dev, sampling_rate, fft_size, win_length, num_mels, fmin, fmax)
    if out is None:
        out = torch.empty((B, Tmax, num_mels), dtype=torch.float32, device=dev)
    seg = _fft8_tables(dev, sampling_rate, fft_size, num_mels, fmin, fmax) if (radix8 is None or radix8) else None
    if radix8 and seg is None:
        raise ValueError("stft_logmel_fft_device: the radix-8 kernel needs n_fft = 1024 and triangular mel filters (n_mels <= 128)")
    if seg is not None:                 # n_fft = 1024: three in-register radix-8 passes, mel by segments (csrc/stft_fft.hip)
        _lib.check(_lib.lib().s2svc_stft_logmel_fft8(B, Nmax, Tmax, hop_size, num_mels, x.data_ptr(), nlen_d.data_ptr(), frames_d.data_ptr(),
                                                     tables.data_ptr(), seg[0].data_ptr(), seg[1].data_ptr(), seg[2].data_ptr(), eps, inv_log,
                                                     None if mean_t is None else mean_t.data_ptr(),
                                                     None if inv_scale_t is None else inv_scale_t.data_ptr(), out.data_ptr(), K.stream()),
                   "stft_logmel_fft8")
        return out
    _lib.check(_lib.lib().s2svc_stft_logmel_fft(B, Nmax, Tmax, fft_size, hop_size, num_mels, x.data_ptr(), nlen_d.data_ptr(),
                                                frames_d.data_ptr(), tables.data_ptr(),
                                                lo.data_ptr(), hi.data_ptr(), off.data_ptr(), melw_n, maxw, eps, inv_log,
                                                None if mean_t is None else mean_t.data_ptr(),
                                                None if inv_scale_t is None else inv_scale_t.data_ptr(), out.data_ptr(), K.stream()),
               "stft_logmel_fft")
    return out


def logmelfilterbank_batch(audios, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                           fmin=None, fmax=None, eps=1e-10, log_base=10.0, mean=None, scale=None, lengths=None, device="cuda",
                           impl=None):
    """B utterances -> (mel (B, Tmax, num_mels) fp32, zero-padded; frames (B,) LongTensor on the host): the batch the
    collater would build from the per-utterance features of bin/preprocess.py + bin/normalize.py, in three launches.

    audios: a list of 1-D float arrays / tensors (any lengths), or a zero-padded (B, Nmax) tensor with `lengths`.
    Every utterance equals `logmelfilterbank(audio_b, ...)` on its first 1 + n_b // hop frames.
    impl: "fft" (ONE launch, FFT in LDS: csrc/stft_fft.hip; n_fft in {512, 1024, 2048}), "gemm" (the DFT-as-GEMM formulation of
    round 2: 3 launches) or None = "fft" where it applies."""
    if window != "hann":
        raise NotImplementedError("only the hann window of the recipes is supported")
    if log_base not in (None, 10.0, 2.0):
        raise ValueError(f"{log_base} is not supported.")
    if isinstance(audios, torch.Tensor) and audios.dim() == 2:
        if lengths is None:
            raise ValueError("a padded (B, Nmax) batch needs `lengths`")
        x = audios.to(device=device, dtype=torch.float32).contiguous()
        nlen = [int(v) for v in lengths]
    else:
        nlen = [int(len(a)) for a in audios]
        host = torch.zeros((len(nlen), max(nlen)), dtype=torch.float32).pin_memory()      # one staging buffer, one H2D copy
        for b, a in enumerate(audios):
            host[b, : nlen[b]] = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float32))
        x = host.to(device, non_blocking=True)
    dev = x.device
    B, Nmax = x.shape
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    pad, nb = fft_size // 2, fft_size // 2 + 1
    frames = [1 + n // hop_size for n in nlen]
    Tmax = max(frames)
    L, st = _lib.lib(), K.stream()
    inv_log = 1.0 if log_base is None else 1.0 / math.log(log_base)
    m_ptr = s_ptr = None
    if mean is not None:
        mean_t = torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous()
        inv_scale_t = (1.0 / torch.as_tensor(scale, dtype=torch.float32, device=dev)).contiguous()
        m_ptr, s_ptr = mean_t.data_ptr(), inv_scale_t.data_ptr()
    nlen_d = torch.tensor(nlen, dtype=torch.int32, device=dev)
    frames_d = torch.tensor(frames, dtype=torch.int32, device=dev)
    if impl not in (None, "fft", "fft_radix4", "gemm"):
        raise ValueError("impl must be None, 'fft', 'fft_radix4' or 'gemm'")
    wl = fft_size if win_length is None else win_length
    fft_ok = fft_size in (512, 1024, 2048) and wl <= fft_size
    if impl in ("fft", "fft_radix4") and not fft_ok:
        raise ValueError("the FFT front-end needs fft_size in {512, 1024, 2048} and win_length <= fft_size")
    if impl != "gemm" and fft_ok:
        out = stft_logmel_fft_device(x, nlen_d, frames_d, Tmax, sampling_rate, fft_size, hop_size, win_length, num_mels, fmin, fmax, eps,
                                     inv_log, mean_t if mean is not None else None, inv_scale_t if mean is not None else None,
                                     radix8=False if impl == "fft_radix4" else None)
        return out, torch.tensor(frames, dtype=torch.long)
    basis, melb = _tables(dev, sampling_rate, fft_size, win_length, num_mels, fmin, fmax)
    key = (str(dev), sampling_rate, fft_size, num_mels, fmin, fmax)
    if key not in _RANGES:
        lo, hi = _mel_ranges(melb.cpu().numpy())
        _RANGES[key] = (torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev))
    lo, hi = _RANGES[key]
    ld = ((Tmax - 1) * hop_size + fft_size + 63) // 64 * 64
    ld = max(ld, Nmax + 2 * pad)
    padded = torch.empty((B, ld), dtype=torch.float32, device=dev)
    _lib.check(L.s2svc_reflect_pad_batch(B, Nmax, pad, ld, x.data_ptr(), nlen_d.data_ptr(), padded.data_ptr(), st), "reflect_pad_batch")
    z = torch.empty((B, Tmax, 2 * nb), dtype=torch.float32, device=dev)
    K.gemm(K.operand(padded, hop_size, bs0=ld), K.operand(basis, fft_size), Tmax, 2 * nb, fft_size, z, in_dtype=torch.float32,
           nb0=B, nb1=1, cbs=(Tmax * 2 * nb, 0))
    out = torch.empty((B, Tmax, num_mels), dtype=torch.float32, device=dev)
    _lib.check(L.s2svc_mel_log_batch(B, Tmax, nb, num_mels, z.data_ptr(), frames_d.data_ptr(), melb.data_ptr(), lo.data_ptr(),
                                     hi.data_ptr(), eps, inv_log, m_ptr, s_ptr, out.data_ptr(), st), "mel_log_batch")
    return out, torch.tensor(frames, dtype=torch.long)
