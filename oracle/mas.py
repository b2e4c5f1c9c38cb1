"""Monotonic alignment search + viterbi_decode, numpy restatement (slow loops; small cases only).

Follows seq2seq_vc/modules/alignments.py:63-93 (_monotonic_alignment_search) and :281-310
(viterbi_decode).  Row 0 is an fp64 running prefix sum (the reference re-sums an fp32 slice per
column; see DESIGN.md section 5, documented deviations).  The C restatement in oracle/mas.c is the fast twin.
"""
import numpy as np


def monotonic_alignment_search(log_p_attn):
    """log_p_attn: (T_mel, T_inp) float32 -> A (T_mel,) int64, min decision margin (float)."""
    lp = np.asarray(log_p_attn, dtype=np.float32)
    T_mel, T_inp = lp.shape
    Q = np.full((T_inp, T_mel), -np.inf, dtype=np.float64)
    acc = 0.0
    for j in range(T_mel):                       # alignments.py:72-73
        acc += float(lp[j, 0])
        Q[0, j] = acc
    for j in range(1, T_mel):                    # alignments.py:76-78
        for i in range(1, min(j + 1, T_inp)):
            Q[i, j] = max(Q[i - 1, j - 1], Q[i, j - 1]) + float(lp[j, i])
    A = np.full((T_mel,), T_inp - 1, dtype=np.int64)
    margin = np.inf
    for j in range(T_mel - 2, -1, -1):           # alignments.py:81-92
        i_b = A[j + 1]
        i_a = i_b - 1
        if i_b == 0:
            a = 0
        else:
            qa, qb = Q[i_a, j], Q[i_b, j]
            if np.isfinite(qa) and np.isfinite(qb):
                margin = min(margin, abs(qa - qb))
            a = i_a if qa >= qb else i_b
        A[j] = a
    return A, margin


def viterbi_decode(log_p_attn, text_lengths, feats_lengths):
    """(B,T_feats,T_text) f32, lens -> ds (B,T_text) f32, bin_loss float, paths list, min margin."""
    lp = np.asarray(log_p_attn, dtype=np.float32)
    B, _, T_text = lp.shape
    ds = np.zeros((B, T_text), dtype=np.float32)
    bin_loss = 0.0
    paths, margin = [], np.inf
    for b in range(B):
        cur = lp[b, : int(feats_lengths[b]), : int(text_lengths[b])]
        A, m = monotonic_alignment_search(cur)
        margin = min(margin, m)
        cnt = np.bincount(A)
        ds[b, : len(cnt)] = cnt
        bin_loss -= float(np.mean(cur[np.arange(cur.shape[0]), A].astype(np.float32)))
        paths.append(A)
    return ds, bin_loss / B, paths, margin
