"""numpy restatement of `logmelfilterbank` (reference bin/preprocess.py:30-92) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference delegates the arithmetic to librosa (setup.cfg:5 `librosa >= 0.8.0`, unpinned,
not installed here) and holds no test vectors for it.  This file restates librosa's documented algorithm
(librosa.stft center=True / reflect padding / periodic Hann; librosa.filters.mel Slaney scale + Slaney area
normalisation) with an FFT-based formulation that is independent of the product's DFT-as-GEMM path.
Cross-checks that exist (tests/test_oracle_golden.py, none of them the reference's own vectors, hence still "unpinned"): torch.stft,
scipy.signal.stft, the published Slaney constants and closed forms, and -- round 6 -- the port of librosa's `stft` / `filters.mel` in
huggingface transformers 5.15 (`transformers.audio_utils`): filter bank equal to 2e-16, log-mel to 2e-8 in float64.
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=float)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    out = f / f_sp
    big = f >= min_log_hz
    out = np.where(big, min_log_mel + np.log(np.where(big, f, min_log_hz) / min_log_hz) / logstep, out)
    return out


def _mel_to_hz(m):
    m = np.asarray(m, dtype=float)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, len(freqs)))
    for i in range(n_mels):
        lo, ce, hi = pts[i], pts[i + 1], pts[i + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        fb[i] = np.clip(np.minimum(up, down), 0, None) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, num_mels=80, fmin=None, fmax=None, eps=1e-10,
                     log_base=10.0, dtype=np.float32):
    """dtype=np.float64 runs the whole chain in double precision (the analytic known-answer tests and the fp32-adequacy
    check of tests/test_oracle_golden.py use it)."""
    if dtype == np.float64:
        return _logmel64(audio, sampling_rate, fft_size, hop_size, num_mels, fmin, fmax, eps, log_base)
    x = np.asarray(audio, dtype=np.float32)
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    xp = np.pad(x, fft_size // 2, mode="reflect")
    n_frames = 1 + len(x) // hop_size
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(fft_size) / fft_size)).astype(np.float32)
    idx = np.arange(fft_size)[None, :] + hop_size * np.arange(n_frames)[:, None]
    spec = np.abs(np.fft.rfft(xp[idx] * win, axis=1)).astype(np.float32)         # (frames, bins)
    mel = np.maximum(eps, spec @ mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax).T)
    if log_base is None:
        return np.log(mel)
    return np.log10(mel) if log_base == 10.0 else np.log2(mel)


def mel_filterbank64(sr, n_fft, n_mels, fmin, fmax):
    """The same triangles in float64, without the final float32 cast."""
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, len(freqs)))
    for i in range(n_mels):
        lo, ce, hi = pts[i], pts[i + 1], pts[i + 2]
        fb[i] = np.clip(np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)), 0, None) * (2.0 / (hi - lo))
    return fb


def _logmel64(audio, sampling_rate, fft_size, hop_size, num_mels, fmin, fmax, eps, log_base):
    x = np.asarray(audio, dtype=np.float64)
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    xp = np.pad(x, fft_size // 2, mode="reflect")
    n_frames = 1 + len(x) // hop_size
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(fft_size) / fft_size)
    idx = np.arange(fft_size)[None, :] + hop_size * np.arange(n_frames)[:, None]
    spec = np.abs(np.fft.rfft(xp[idx] * win, axis=1))
    mel = np.maximum(eps, spec @ mel_filterbank64(sampling_rate, fft_size, num_mels, fmin, fmax).T)
    if log_base is None:
        return np.log(mel)
    return np.log10(mel) if log_base == 10.0 else np.log2(mel)
