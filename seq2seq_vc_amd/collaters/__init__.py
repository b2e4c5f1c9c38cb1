"""Batch collaters resolved by name (`collater_type:`), mirroring `seq2seq_vc.collaters`
(reference collaters/ar_vc.py:11-73, nar_vc.py:12-91, ar_tts.py:11-64): zero-pad every utterance to the
batch maximum and return lengths as CPU LongTensors (the models consume lengths on the host to size kernels)."""
import numpy as np
import torch


def pad_batch(seqs, dtype, pad_value=0):
    """[(T_i, *)] numpy arrays -> (B, T_max, *) tensor; one allocation, one copy per utterance."""
    tmax = max(s.shape[0] for s in seqs)
    out = torch.full((len(seqs), tmax) + tuple(seqs[0].shape[1:]), pad_value, dtype=dtype)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = torch.as_tensor(np.asarray(s)).to(dtype)
    return out


def _lens(seqs):
    return torch.tensor([s.shape[0] for s in seqs], dtype=torch.long)


def stop_labels(olens, tmax):
    """1.0 from the last valid frame on (ar_vc.py:60-62)."""
    return (torch.arange(tmax)[None, :] >= (olens[:, None] - 1)).float()


class ARVCCollater(object):
    def __call__(self, batch):
        xs, ys = [b["src_feat"] for b in batch], [b["trg_feat"] for b in batch]
        ilens, olens = _lens(xs), _lens(ys)
        xs, ys = pad_batch(xs, torch.float32), pad_batch(ys, torch.float32)
        return {"xs": xs, "ilens": ilens, "ys": ys, "olens": olens, "labels": stop_labels(olens, ys.size(1)), "spembs": None}


class NARVCCollater(object):
    def __call__(self, batch):
        xs, ys, dps = [b["src_feat"] for b in batch], [b["trg_feat"] for b in batch], [b["dp_input"] for b in batch]
        items = {"xs": pad_batch(xs, torch.float32), "ilens": _lens(xs), "ys": pad_batch(ys, torch.float32), "olens": _lens(ys),
                 "dp_inputs": pad_batch(dps, torch.float32), "dplens": _lens(dps), "spembs": None}
        if "duration" in batch[0]:
            ds = [b["duration"] for b in batch]
            items["durations"] = pad_batch(ds, torch.long)
            # (the reference measures the lengths AFTER padding, nar_vc.py:84-88: every entry is the padded length)
            items["duration_lens"] = torch.full((len(ds),), items["durations"].shape[1], dtype=torch.long)
        return items


class ARTTSCollater(object):
    def __call__(self, batch):
        xs, ys = [b[0] for b in batch], [b[1] for b in batch]
        ilens, olens = _lens(xs), _lens(ys)
        xs, ys = pad_batch(xs, torch.long), pad_batch(ys, torch.float32)
        return xs, ilens, ys, stop_labels(olens, ys.size(1)), olens, None
