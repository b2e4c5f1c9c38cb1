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


class _DeviceBatch(object):
    """Builds padded batches ON THE DEVICE: the utterances of a batch are concatenated into one pinned host buffer (a single
    H2D copy for the whole batch instead of a padded tensor per field), and one kernel (csrc/frontend.hip
    `ragged_to_padded`) scatters them into the zero-padded (B, Tmax, D) layout, applies the optional mean/variance
    normalisation of bin/normalize.py:172-193 and writes the stop labels.  Lengths stay host LongTensors (the models size
    kernels with them).  `sort_by_length=True` orders a batch longest-first (bucketing is the sampler's business; within a
    batch the order only matters to RNN-style consumers and is off by default, as in the reference)."""

    def __init__(self, device="cuda", mean=None, scale=None, sort_by_length=False, stats=None):
        """mean / scale: statistics applied to every field (one stats file for source and target, as when both sides share a
        feature extractor).  stats: {"src": (mean, scale), "trg": (...), "dp": (...)} -- per-field statistics as the reference's
        recipes keep them (separate source / target stats files, bin/normalize.py:172-193); a field without an entry is not
        normalised.  The feature dimension of a field must equal the length of its statistics."""
        self.device = torch.device(device)
        self.sort_by_length = sort_by_length
        dev_pair = lambda m, sc: (torch.as_tensor(m, dtype=torch.float32, device=self.device).contiguous(),
                                  (1.0 / torch.as_tensor(sc, dtype=torch.float32, device=self.device)).contiguous())
        self.stats = {}
        if mean is not None:
            if scale is None:
                raise ValueError("device collater: `mean` needs `scale`")
            self.stats = {k: dev_pair(mean, scale) for k in ("src", "trg", "dp")}
        for k, (m, sc) in (stats or {}).items():
            if k not in ("src", "trg", "dp"):
                raise ValueError(f"device collater: unknown statistics field '{k}' (src / trg / dp)")
            self.stats[k] = dev_pair(m, sc)

    def _pad(self, seqs, want_labels=False, field=None):
        from .. import _lib
        from ..ops import kernels as K
        lens = [int(s.shape[0]) for s in seqs]
        D = int(seqs[0].shape[1])
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        host = torch.empty((int(off[-1]), D), dtype=torch.float32).pin_memory()
        for i, s in enumerate(seqs):
            host[off[i]: off[i + 1]] = torch.as_tensor(np.asarray(s), dtype=torch.float32)
        ragged = host.to(self.device, non_blocking=True)
        offs = torch.from_numpy(off).to(self.device, non_blocking=True)
        B, Tmax = len(seqs), max(lens)
        out = torch.empty((B, Tmax, D), dtype=torch.float32, device=self.device)
        labels = torch.empty((B, Tmax), dtype=torch.float32, device=self.device) if want_labels else None
        st = self.stats.get(field)
        if st is not None and (st[0].numel() != D or st[1].numel() != D):
            raise ValueError(f"device collater: '{field}' features have {D} dimensions but their statistics {st[0].numel()}")
        _lib.check(_lib.lib().s2svc_ragged_to_padded(B, Tmax, D, ragged.data_ptr(), offs.data_ptr(),
                                                     st[0].data_ptr() if st is not None else None,
                                                     st[1].data_ptr() if st is not None else None, out.data_ptr(),
                                                     None if labels is None else labels.data_ptr(), K.stream()), "ragged_to_padded")
        return out, torch.tensor(lens, dtype=torch.long), labels

    def _order(self, batch, key):
        if not self.sort_by_length:
            return batch
        return sorted(batch, key=lambda b: -b[key].shape[0])


class DeviceARVCCollater(_DeviceBatch):
    """ARVCCollater (collaters/ar_vc.py:11-73) with the batch assembled on the device."""

    def __call__(self, batch):
        batch = self._order(batch, "src_feat")
        xs, ilens, _ = self._pad([b["src_feat"] for b in batch], field="src")
        ys, olens, labels = self._pad([b["trg_feat"] for b in batch], want_labels=True, field="trg")
        return {"xs": xs, "ilens": ilens, "ys": ys, "olens": olens, "labels": labels, "spembs": None}


class DeviceNARVCCollater(_DeviceBatch):
    """NARVCCollater (collaters/nar_vc.py:12-91) with the batch assembled on the device."""

    def __call__(self, batch):
        batch = self._order(batch, "src_feat")
        xs, ilens, _ = self._pad([b["src_feat"] for b in batch], field="src")
        ys, olens, _ = self._pad([b["trg_feat"] for b in batch], field="trg")
        dps, dplens, _ = self._pad([b["dp_input"] for b in batch], field="dp")
        items = {"xs": xs, "ilens": ilens, "ys": ys, "olens": olens, "dp_inputs": dps, "dplens": dplens, "spembs": None}
        if "duration" in batch[0]:
            ds = [b["duration"] for b in batch]
            items["durations"] = pad_batch(ds, torch.long)          # consumed on the host (output sizes), stays there
            items["duration_lens"] = torch.full((len(ds),), items["durations"].shape[1], dtype=torch.long)
        return items
