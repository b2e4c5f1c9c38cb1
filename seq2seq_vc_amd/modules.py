"""Building blocks with the reference's class names, constructor arguments and state_dict keys,
whose forward passes run on the HIP kernels (ops/functional.py).

torch.nn.Linear / Conv1d / Conv2d / BatchNorm1d / LayerNorm / Embedding objects are used purely as
PARAMETER HOLDERS (so keys, shapes, buffers and default initialisation equal the reference's); their
own forward() is never called.  Sequence masks are replaced by `Lens` objects (host tuple + cached
int32 device vector) -- no (B,1,T) / (B,T,T) bool tensors are ever built.

Reference files: seq2seq_vc/modules/transformer/*.py, modules/conformer/*.py, modules/pre_postnets.py,
layers/positional_encoding.py.
"""
import math
import os

import torch
from torch import nn

from .ops import functional as Fn
from .ops import kernels as K

# ------------------------------------------------------------------------------------------------
# lengths instead of masks
# ------------------------------------------------------------------------------------------------
_LENS_CACHE = {}
_BANK = None          # the LensBank of the training step being captured / traced (trainers: config["hip_graph"])


class LensBank:
    """Lengths as DATA of a captured training step.  A hipGraph bakes in what the host computed while it was captured: with
    the by-value cache below, the int32 device vectors of one batch's lengths.  While a bank is active every `Lens` gets a
    SLOT of one persistent device buffer instead, and remembers how it was derived (a root registered by the trainer, or
    `map` / `clamp` of another Lens).  Before a replay `refresh()` recomputes every slot on the host from the new batch's
    root lengths -- the same lambdas, in creation order -- and ships them with ONE copy; the kernels of the graph read the
    slots.  Host-side uses of the lengths must not differ between batches of one graph: `max()` of a banked Lens is its
    `cap` (the padded length of the tensor it describes), and lengths that reach a step without provenance raise instead of
    being baked in.

    The reference crops a batch to its longest utterance before it computes (models/vtn.py:208-214, 269-271; the collater hands
    models/aas_vc.py a batch padded to exactly that length), a captured step cannot (shapes are baked in).  So a banked Lens also
    carries `ext`: the length the reference's CROPPED tensor has -- max of the root lengths, pushed through the same arithmetic as
    the tensor shapes (`map(fn, ext_fn)`).  Frames between `ext` and `cap` are ABSENT: `Lens.crop()` is the slot (B copies of
    `ext`, graph data like every other slot) that the kernels which mix along time or over the batch take as `vlens` -- Conv1d
    with k > 1 (ops.functional.conv1d / crop_rows), BatchNorm (batch_norm_act), the Conformer convolution module
    (functional_aas.convmod_core), nearest-neighbour resampling -- so that a captured step on a batch that does not fill its padded
    shape computes exactly what the reference computes on the cropped batch."""

    def __init__(self, device, max_slots=96):
        self.device = torch.device(device)
        self.max_slots = max_slots
        self.B = None
        self.buf = None               # (max_slots, B) int32 on the device
        self.entries = []             # creation order: (lens, source): ("root", name) | ("map", parent, fn, ext_fn, with_ext) | ("crop", parent)
        self.roots = {}               # id(tensor the trainer registered) -> Lens
        self.closed = False           # after the capture: no new slots

    def root(self, name, source, values, cap):
        """Register the lengths `values` (B ints) the trainer got as `source` (the very object it hands to the model)."""
        lens = Lens.__new__(Lens)
        lens._init_banked(self, values, cap, ("root", name))
        self.roots[id(source)] = (source, lens)
        return lens

    def _slot(self, lens, source):
        if self.closed:
            raise RuntimeError("LensBank: a Lens was created after the capture of this step finished")
        k = len(self.entries)
        B = len(lens.host)
        if self.buf is None:
            self.B = B
            self.buf = torch.zeros((self.max_slots, B), dtype=torch.int32, device=self.device)
        if B != self.B or k >= self.max_slots:
            raise RuntimeError(f"LensBank: {B} lengths in a bank of batch size {self.B}, or more than {self.max_slots} slots")
        self.entries.append((lens, source))
        if not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # eager ("traced") step: the kernels run now.  (.data: a slot written while the forward pass is under way must not bump the
            # version counter that autograd checks on the slots it has already saved for the backward pass)
            self.buf.data[k].copy_(torch.tensor(lens.host, dtype=torch.int32))
        return self.buf[k]

    def upload(self):
        """Ship the host values of every slot: one copy from a FRESH pinned block (the host may run several replays ahead of the
        device; torch's pinned allocator does not hand a block out again while a copy from it is pending)."""
        n = len(self.entries)
        if n:
            stage = torch.tensor([lens.host for lens, _ in self.entries], dtype=torch.int32)
            if self.device.type == "cuda":
                stage = stage.pin_memory()
            self.buf.data[:n].copy_(stage, non_blocking=True)

    def refresh(self, root_values):
        """root_values: name -> B ints of the new batch.  Recomputes every slot and uploads them (current stream)."""
        for k, (lens, source) in enumerate(self.entries):          # creation order: a parent precedes what was derived from it
            if source[0] == "root":
                vals = tuple(int(v) for v in root_values[source[1]])
                if max(vals) > lens.cap:
                    raise ValueError(f"LensBank: length {max(vals)} of '{source[1]}' exceeds the padded length {lens.cap} of this graph")
                ext = max(vals)
            else:
                vals, ext = _derive(source, len(lens.host))
            lens.host, lens.ext = vals, ext
        self.upload()


def _derive(source, B):
    """(values, ext) of a banked Lens from its parent's CURRENT values: ("map", parent, fn, ext_fn, with_ext) | ("crop", parent)."""
    parent = source[1]
    if source[0] == "crop":
        return (parent.ext,) * B, parent.ext
    _, _, fn, ext_fn, with_ext = source
    vals = tuple(fn(v, parent.ext) for v in parent.host) if with_ext else tuple(fn(v) for v in parent.host)
    return vals, (ext_fn if ext_fn is not None else fn)(parent.ext)


class lens_bank:
    """with lens_bank(bank): ... -- Lens objects created inside belong to `bank`."""

    def __init__(self, bank):
        self.bank = bank

    def __enter__(self):
        global _BANK
        self.prev, _BANK = _BANK, self.bank
        return self.bank

    def __exit__(self, *exc):
        global _BANK
        _BANK = self.prev


def tag_lens(tensor, lens):
    """Attach the Lens a host-side length tensor was made from (model outputs such as `olens` after the reduction-factor
    trim): `Lens.of` returns it instead of reading the tensor's values, so the provenance survives the trip through the
    caller (model -> trainer -> criterion)."""
    if _BANK is not None and isinstance(tensor, torch.Tensor):
        tensor._s2s_lens = lens
    return tensor


class Lens:
    """Valid lengths of a padded batch: `.host` tuple of ints, `.dev` int32 device tensor (cached by
    value so that steady-state steps issue no H2D copies and stay hipGraph-capturable; a slot of the active LensBank
    when a training step is captured with lengths as data)."""

    def __init__(self, values, device, _source=None, _cap=None, _ext=None):
        if _BANK is not None:
            if _source is None:
                raise RuntimeError("Lens: lengths without provenance inside a captured training step (register them with "
                                   "LensBank.root, derive them with Lens.map / Lens.clamp, or tag host tensors with tag_lens)")
            self._init_banked(_BANK, values, _cap, _source, _ext)
            return
        self.host = tuple(int(v) for v in values)
        self.device = torch.device(device)
        self.cap = None
        self.ext = None               # no bank: the models crop the batch themselves, every row of a tensor is present
        key = (self.host, self.device.type, self.device.index)
        t = _LENS_CACHE.get(key)
        if t is None:
            if len(_LENS_CACHE) > 4096:
                _LENS_CACHE.clear()
            t = torch.tensor(self.host, dtype=torch.int32, device=self.device)
            _LENS_CACHE[key] = t
        self.dev = t

    def _init_banked(self, bank, values, cap, source, ext=None):
        self.host = tuple(int(v) for v in values)
        self.device = bank.device
        self.cap = int(cap)
        self.ext = int(max(self.host) if ext is None else ext)      # a root: the reference crops to the longest utterance
        self._crop = None
        self.dev = bank._slot(self, source)

    @staticmethod
    def of(lens, device):
        if lens is None or isinstance(lens, Lens):
            return lens
        if _BANK is not None:
            tagged = getattr(lens, "_s2s_lens", None)
            if tagged is not None:
                return tagged
            reg = _BANK.roots.get(id(lens))
            if reg is not None and reg[0] is lens:
                return reg[1]
            raise RuntimeError("Lens.of: these lengths were not registered with the LensBank of the captured step")
        if isinstance(lens, torch.Tensor):
            lens = lens.tolist()
        return Lens(lens, device)

    def map(self, fn, ext_fn=None, with_ext=False):
        """Lengths fn(v) of a tensor derived from the one these lengths describe.  ext_fn: how the derived tensor's time axis
        follows from this one's when that is not fn itself (Conv2dSubsampling: lengths ceil(v / 4), axis ((T - 1) // 2 - 1) // 2);
        with_ext: fn(v, ext) also sees the cropped length of THIS tensor (banked lengths; else ext = what max() returns)."""
        if self.cap is not None:
            vals, ext = _derive(("map", self, fn, ext_fn, with_ext), len(self.host))
            return Lens(vals, self.device, _source=("map", self, fn, ext_fn, with_ext), _cap=(ext_fn if ext_fn is not None else fn)(self.cap),
                        _ext=ext)
        if with_ext:
            return Lens([fn(v, self.max()) for v in self.host], self.device)
        return Lens([fn(v) for v in self.host], self.device)

    def crop(self):
        """The `vlens` of the kernels that must not see the frames between the reference's cropped length and the padded length
        (LensBank): B copies of `ext` in a slot of the bank.  None outside a bank (every frame of a tensor is present)."""
        if self.cap is None:
            return None
        if self._crop is None:
            self._crop = Lens((self.ext,) * len(self.host), self.device, _source=("crop", self), _cap=self.cap, _ext=self.ext)
        return self._crop

    def max(self):
        return self.cap if self.cap is not None else max(self.host)

    def clamp(self, hi):
        return self.map(lambda v, _hi=hi: min(v, _hi))


def crop_dev(lens):
    """Device vector for the `vlens` argument of the time-mixing kernels, or None (no bank / no lengths)."""
    if lens is None or lens.cap is None:
        return None
    return lens.crop().dev


# ------------------------------------------------------------------------------------------------
# positional encodings (layers/positional_encoding.py)
# ------------------------------------------------------------------------------------------------
def _sin_table(n, d, reverse=False):
    pos = torch.arange(n - 1, -1, -1.0, dtype=torch.float32) if reverse else torch.arange(0, n, dtype=torch.float32)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos[:, None] * div)
    pe[:, 1::2] = torch.cos(pos[:, None] * div)
    return pe


class PositionalEncoding(nn.Module):
    """x*sqrt(d) + pe[:T]  (positional_encoding.py:14-70).  The fp32 table is a lazily built,
    non-persistent attribute exactly as in the reference (not part of the state_dict)."""

    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model, self.reverse, self.max_len = d_model, reverse, max_len
        self.xscale = math.sqrt(d_model)
        self.dropout_rate = dropout_rate
        self._pe = None

    def table(self, T, device):
        if self._pe is None or self._pe.shape[0] < T or self._pe.device != device:
            self._pe = _sin_table(max(self.max_len, T), self.d_model, self.reverse).to(device)
        return self._pe

    def forward(self, x):
        p = self.dropout_rate if self.training else 0.0
        return Fn.posenc(x, self.table(x.shape[1], x.device), None, self.xscale, p)


class ScaledPositionalEncoding(PositionalEncoding):
    """x + alpha*pe[:T]  (positional_encoding.py:73-106)."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model, dropout_rate, max_len)
        self.alpha = nn.Parameter(torch.tensor(1.0))

    def reset_parameters(self):
        self.alpha.data = torch.tensor(1.0)

    def forward(self, x):
        p = self.dropout_rate if self.training else 0.0
        return Fn.posenc(x, self.table(x.shape[1], x.device), self.alpha, 1.0, p)


class RelPositionalEncoding(nn.Module):
    """(x*sqrt(d), pos_emb (1, 2T-1, d))  (positional_encoding.py:238-309)."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model, self.dropout_rate = d_model, dropout_rate
        self.xscale = math.sqrt(d_model)
        self._cache = {}

    def pos_emb(self, T, device, dtype):
        key = (T, str(device), dtype)
        if key not in self._cache:
            if len(self._cache) > 64:
                self._cache.clear()
            plus = torch.flip(_sin_table(T, self.d_model), [0])
            pos = torch.arange(0, T, dtype=torch.float32)[:, None]
            div = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d_model))
            minus = torch.zeros(T, self.d_model)
            minus[:, 0::2], minus[:, 1::2] = torch.sin(-pos * div), torch.cos(-pos * div)
            self._cache[key] = torch.cat([plus, minus[1:]], dim=0)[None].to(device).to(dtype).contiguous()
        return self._cache[key]

    def forward(self, x):
        p = self.dropout_rate if self.training else 0.0
        pe = self.pos_emb(x.shape[1], x.device, x.dtype)
        return Fn.posenc(x, None, None, self.xscale, p), Fn.dropout(pe, p)


class LegacyRelPositionalEncoding(PositionalEncoding):
    """(x*sqrt(d), first T rows of the reversed max_len table)  (positional_encoding.py:198-235)."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model, dropout_rate, max_len, reverse=True)

    def forward(self, x):
        if _BANK is not None:
            # the legacy table is sliced by the tensor's length (rows max_len - T .. max_len - 1 reversed) and the legacy rel_shift
            # wraps rows at that length: both depend on the CROPPED length, which a captured step only has as data
            raise NotImplementedError('config["hip_graph"]: legacy relative positional encoding is not supported in captured steps '
                                      "(no recipe uses it); run this model with hip_graph off")
        p = self.dropout_rate if self.training else 0.0
        pe = K.cast(self.table(x.shape[1], x.device)[: x.shape[1]][None].contiguous(), x.dtype)
        return Fn.posenc(x, None, None, self.xscale, p), Fn.dropout(pe, p)


# ------------------------------------------------------------------------------------------------
# attention (modules/transformer/attention.py)
# ------------------------------------------------------------------------------------------------
class MultiHeadedAttention(nn.Module):
    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.attn = None
        self.dropout_rate = dropout_rate

    def _p(self):
        return self.dropout_rate if self.training else 0.0

    def forward(self, query, key, value, klens=None, causal=False, kv=None, passthrough=False):
        """klens: Lens of valid key positions (None = all valid); causal adds j<=i.
        kv: the packed K/V projection of `key` (= `value`) computed by the caller (decoder stacks project the memory
        for all their layers with ONE GEMM).
        passthrough=True (only with the packed projections): returns (out, alias of query) -- a post-LN layer takes its
        residual from the alias, whose gradient then rides in the query projection's data-gradient GEMM (Fn.linear)."""
        kl = None if klens is None else klens.dev
        f = getattr(self, "_fused", None)   # packed Q/K/V views of the flat parameter buffer (optim.FlatAdam)
        qp = None
        if kv is not None:
            q = Fn.linear(query, f["w_q"], f["b_q"], passthrough=passthrough)
            if passthrough:
                q, qp = q
            ctx, self.attn = Fn.attention_packed_kv(q, kv, kl, causal, self.h, self._p())
        elif f is not None and key is value and (query is not key or "w_qkv" in f):
            if query is key:
                qkv = Fn.linear(query, f["w_qkv"], f["b_qkv"], passthrough=passthrough)   # ONE GEMM, N = 3D
                if passthrough:
                    qkv, qp = qkv
                ctx, self.attn = Fn.attention_packed_qkv(qkv, kl, causal, self.h, self._p())
            else:
                q = Fn.linear(query, f["w_q"], f["b_q"], passthrough=passthrough)
                if passthrough:
                    q, qp = q
                kv = Fn.linear(key, f["w_kv"], f["b_kv"])                           # ONE GEMM, N = 2D
                ctx, self.attn = Fn.attention_packed_kv(q, kv, kl, causal, self.h, self._p())
        else:
            q = Fn.linear(query, self.linear_q.weight, self.linear_q.bias)
            k = Fn.linear(key, self.linear_k.weight, self.linear_k.bias)
            v = Fn.linear(value, self.linear_v.weight, self.linear_v.bias)
            ctx, self.attn = Fn.attention_core(q, k, v, kl, causal, self.h, self._p())
            qp = query
        out = Fn.linear(ctx, self.linear_out.weight, self.linear_out.bias)
        return (out, qp) if passthrough else out


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    """attention.py:209-305 (new rel_shift); `legacy=True` gives attention.py:114-206."""

    legacy = False

    def __init__(self, n_head, n_feat, dropout_rate, zero_triu=False):
        super().__init__(n_head, n_feat, dropout_rate)
        if zero_triu:
            raise NotImplementedError("zero_triu is not supported")
        self.zero_triu = zero_triu
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def forward(self, query, key, value, pos_emb, klens=None):
        f = getattr(self, "_fused", None)   # packed Q/K/V views of the flat parameter buffer (optim.FlatAdam)
        if f is not None and "w_qkv" in f and query is key and key is value:
            qkv = Fn.linear(query, f["w_qkv"], f["b_qkv"])                              # ONE GEMM, N = 3D
            pos = Fn.linear(pos_emb, self.linear_pos.weight, None)
            ctx, self.attn = Fn.rel_attention_packed(qkv, pos, self.pos_bias_u, self.pos_bias_v,
                                                     None if klens is None else klens.dev, self.h, self._p(),
                                                     2 if self.legacy else 1)
            return Fn.linear(ctx, self.linear_out.weight, self.linear_out.bias)
        q = Fn.linear(query, self.linear_q.weight, self.linear_q.bias)
        k = Fn.linear(key, self.linear_k.weight, self.linear_k.bias)
        v = Fn.linear(value, self.linear_v.weight, self.linear_v.bias)
        pos = Fn.linear(pos_emb, self.linear_pos.weight, None)
        qu, qv = Fn.add_head_bias(q, self.pos_bias_u, self.pos_bias_v)
        ctx, self.attn = Fn.rel_attention_core(qu, qv, k, v, pos, None if klens is None else klens.dev, self.h, self._p(),
                                               2 if self.legacy else 1)
        return Fn.linear(ctx, self.linear_out.weight, self.linear_out.bias)


class LegacyRelPositionMultiHeadedAttention(RelPositionMultiHeadedAttention):
    legacy = True


# ------------------------------------------------------------------------------------------------
# feed-forward
# ------------------------------------------------------------------------------------------------
class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(act(w_1 x)))  (positionwise_feed_forward.py:12-32); act 'relu' or 'swish'."""

    def __init__(self, idim, hidden_units, dropout_rate, activation="relu"):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden_units)
        self.w_2 = nn.Linear(hidden_units, idim)
        self.dropout_rate = dropout_rate
        self.activation = activation

    def forward(self, x, passthrough=False):
        """passthrough=True -> (y, alias of x) for a post-LN residual (Fn.linear)."""
        p = self.dropout_rate if self.training else 0.0
        if self.activation in ("relu", "swish"):   # activation + dropout (and their derivatives) in the GEMM epilogues
            return Fn.ffn_act(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, self.activation, p, passthrough)
        xp = None
        if passthrough:
            h, xp = Fn.linear(x, self.w_1.weight, self.w_1.bias, passthrough=True)
        else:
            h = Fn.linear(x, self.w_1.weight, self.w_1.bias)
        y = Fn.linear(Fn.act_dropout(h, self.activation, p), self.w_2.weight, self.w_2.bias)
        return (y, xp) if passthrough else y


class MultiLayeredConv1d(nn.Module):
    """Conv1d-ReLU-dropout-Conv1d FFN (multi_layer_conv.py:12-63), channel-last in and out."""

    def __init__(self, in_chans, hidden_chans, kernel_size, dropout_rate):
        super().__init__()
        self.w_1 = nn.Conv1d(in_chans, hidden_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.w_2 = nn.Conv1d(hidden_chans, in_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.dropout_rate = dropout_rate

    def forward(self, x, lens=None):
        """lens: the Lens of x -- in a captured step its crop() keeps the frames the reference does not have out of the taps."""
        p = self.dropout_rate if self.training else 0.0
        vl = crop_dev(lens)
        h = Fn.dropout(Fn.conv1d(x, self.w_1.weight, self.w_1.bias, act="relu", vlens=vl), p)
        return Fn.conv1d(h, self.w_2.weight, self.w_2.bias, vlens=vl)


class LayerNorm(nn.LayerNorm):
    """Parameter holder for LayerNorm(eps=1e-12) (layer_norm.py:12-42)."""

    def __init__(self, nout, dim=-1, eps=1e-12):
        super().__init__(nout, eps=eps)
        self.dim = dim

    def forward(self, x):
        return Fn.layer_norm(x, self.weight, self.bias, self.eps)


def _sub_pass(sub, x, *args, **kw):
    """sublayer(x, ...) -> (output, x for the residual).  MultiHeadedAttention / PositionwiseFeedForward hand back a
    pass-through alias of x (their first GEMM's backward absorbs the residual gradient); other sublayers get x itself."""
    if (type(sub) in (MultiHeadedAttention, PositionwiseFeedForward) and torch.is_grad_enabled() and x.requires_grad):
        return sub(x, *args, passthrough=True, **kw)
    return sub(x, *args, **kw), x


def _res_norm(norm, res, h, p, hscale=1.0):
    """s = res + hscale*dropout(h); returns (LayerNorm(s), s)."""
    return Fn.add_dropout_layer_norm(res, h, norm.weight, norm.bias, norm.eps, p, hscale)


# ------------------------------------------------------------------------------------------------
# Transformer layers (encoder_layer.py:61-119, decoder_layer.py:63-134); concat_after unsupported
# ------------------------------------------------------------------------------------------------
class EncoderLayer(nn.Module):
    def __init__(self, size, self_attn, feed_forward, dropout_rate, normalize_before=True, concat_after=False,
                 stochastic_depth_rate=0.0):
        super().__init__()
        if concat_after or stochastic_depth_rate > 0:
            raise NotImplementedError("concat_after / stochastic depth are not used by any recipe")
        self.self_attn, self.feed_forward = self_attn, feed_forward
        self.norm1, self.norm2 = LayerNorm(size), LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.size, self.normalize_before = size, normalize_before

    def forward(self, x, klens, normed=None):
        """Pre-LN: takes (x, LN1(x)) and returns (x_out, None); the caller chains the next norm.
        Implemented as explicit residual/norm steps so every add+dropout+LN is ONE fused kernel."""
        p = self.dropout_rate if self.training else 0.0
        if self.normalize_before:
            if normed is None:        # x feeds the norm AND the residual: the residual takes the norm's pass-through alias
                y, x = Fn.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, passthrough=True)
            else:
                y = normed
            a = self.self_attn(y, y, y, klens)
            y2, x = _res_norm(self.norm2, x, a, p)
            f = self.feed_forward(y2, klens) if isinstance(self.feed_forward, MultiLayeredConv1d) else self.feed_forward(y2)
            return x, f  # caller adds f with dropout (fused into the next LayerNorm)
        # post-LN: x feeds the sublayer AND the residual; the residual takes it from the sublayer's pass-through alias so
        # that the two gradients meet inside the first data-gradient GEMM instead of in an element-wise add
        a, xr = _sub_pass(self.self_attn, x, x, x, klens)
        x, _ = _res_norm(self.norm1, xr, a, p)
        if isinstance(self.feed_forward, MultiLayeredConv1d):
            f, xr = self.feed_forward(x, klens), x
        else:
            f, xr = _sub_pass(self.feed_forward, x)
        x, _ = _res_norm(self.norm2, xr, f, p)
        return x, None


class DecoderLayer(nn.Module):
    def __init__(self, size, self_attn, src_attn, feed_forward, dropout_rate, normalize_before=True, concat_after=False):
        super().__init__()
        if concat_after:
            raise NotImplementedError("concat_after is not used by any recipe")
        self.size = size
        self.self_attn, self.src_attn, self.feed_forward = self_attn, src_attn, feed_forward
        self.norm1, self.norm2, self.norm3 = LayerNorm(size), LayerNorm(size), LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.normalize_before = normalize_before

    def self_block(self, x, tgt_lens, normed=None, causal=True):
        """The part of the layer that does not see the memory: the self-attention sub-layer up to the input of the
        source attention.  -> (y, x): y feeds the source attention, x is the residual stream (post-LN: the same tensor)."""
        p = self.dropout_rate if self.training else 0.0
        if self.normalize_before:
            if normed is None:
                y, x = Fn.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, passthrough=True)
            else:
                y = normed
            a = self.self_attn(y, y, y, tgt_lens, causal=causal)
            return _res_norm(self.norm2, x, a, p)
        a, xr = _sub_pass(self.self_attn, x, x, x, tgt_lens, causal=causal)
        x, _ = _res_norm(self.norm1, xr, a, p)
        return x, x

    def forward(self, x, tgt_lens, memory, mem_lens, normed=None, causal=True, kv=None, head=None):
        """head: the result of self_block() when the caller ran it ahead of the encoder (Decoder.head)."""
        p = self.dropout_rate if self.training else 0.0
        if self.normalize_before:
            y, x = self.self_block(x, tgt_lens, normed, causal) if head is None else head
            a = self.src_attn(y, memory, memory, mem_lens, kv=kv)
            y, x = _res_norm(self.norm3, x, a, p)
            return x, self.feed_forward(y)
        x = self.self_block(x, tgt_lens, None, causal)[0] if head is None else head[0]
        a, xr = _sub_pass(self.src_attn, x, memory, memory, mem_lens, kv=kv)
        x, _ = _res_norm(self.norm2, xr, a, p)
        f, xr = _sub_pass(self.feed_forward, x)
        x, _ = _res_norm(self.norm3, xr, f, p)
        return x, None


def run_stack(layers, x, after_norm, pre_ln, dropout_rate, training, *args, per_layer=None, **kw):
    """Run a stack of EncoderLayer/DecoderLayer.  In pre-LN mode each layer returns (x, pending FFN
    output); `x + dropout(f)` is fused into the NEXT layer's first LayerNorm (or after_norm).
    per_layer: optional list of extra keyword arguments, one dict per layer."""
    p = dropout_rate if training else 0.0
    pending = None
    cut_name = kw.pop("cut_name", None)      # data-parallel overlap: "<name>.<i>" cuts the graph at the input of layer i
    for li, layer in enumerate(layers):
        kwl = dict(kw, **per_layer[li]) if per_layer is not None else kw      # this layer's own extras only
        if cut_name is not None and li > 0:
            x, pending = Fn.cut_point((x, pending), f"{cut_name}.{li}")
        if pre_ln and pending is not None:
            normed, x = _res_norm(layer.norm1, x, pending, p)
            x, pending = layer(x, *args, normed=normed, **kwl)
        else:
            x, pending = layer(x, *args, **kwl)
    if pre_ln:
        if pending is not None:
            y, x = _res_norm(after_norm, x, pending, p)
            return y
        return after_norm(x)
    return x


# ------------------------------------------------------------------------------------------------
# Conv2d subsampling front-end (subsampling.py:44-105)
# ------------------------------------------------------------------------------------------------
class Conv2dSubsampling(nn.Module):
    def __init__(self, idim, odim, dropout_rate, pos_enc=None, use_pos_enc=True):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, odim, 3, 2), nn.ReLU(), nn.Conv2d(odim, odim, 3, 2), nn.ReLU())
        self.f2 = ((idim - 1) // 2 - 1) // 2
        if use_pos_enc:
            self.out = nn.Sequential(nn.Linear(odim * self.f2, odim),
                                     pos_enc if pos_enc is not None else PositionalEncoding(odim, dropout_rate))
        else:
            self.out = nn.Linear(odim * self.f2, odim)
        self.odim, self.use_pos_enc = odim, use_pos_enc

    def __getitem__(self, key):
        if key != -1:
            raise NotImplementedError("Support only `-1` (for `reset_parameters`).")
        return self.out[key]

    @staticmethod
    def out_lens(lens, t_out, exact=False):
        """mask[:, :, :-2:2][:, :, :-2:2] of a non-pad mask keeps frame t' iff 4*t' < len (the reference's
        batch semantics: frames whose receptive field reaches into the padding stay valid).  exact=True:
        the frame count the utterance has when it is processed alone, ((len-1)//2-1)//2."""
        if lens is None:
            return None
        sub = lambda T: ((T - 1) // 2 - 1) // 2          # noqa: E731  -- frames of the subsampled axis
        if exact:
            return lens.map(lambda v: min(sub(v), t_out), ext_fn=sub)
        if lens.cap is not None:
            # captured step: the reference's mask has sub(longest utterance) frames, not sub(padded length)
            assert sub(lens.cap) == t_out
            return lens.map(lambda v, e: min((v + 3) // 4, sub(e)), ext_fn=sub, with_ext=True)
        return lens.map(lambda v: min((v + 3) // 4, t_out))

    def forward(self, x, lens, exact_lens=False):
        B, T, idim = x.shape
        c0, c2 = self.conv[0], self.conv[2]
        if c0.weight.shape[0] % 8 == 0:
            y = Fn.conv_in1_relu(x, c0.weight, c0.bias, grad_premasked=True)   # direct streaming kernel (C_in = 1)
            first_relu = True
        else:
            y = Fn.conv2d_s2_relu(x.reshape(B, T, idim, 1), c0.weight, c0.bias)
            first_relu = False
        # (B, T2, F2, C) channel-last; relu' of the first layer rides in this layer's data-gradient epilogue
        y = Fn.conv2d_s2_relu(y, c2.weight, c2.bias, grad_premasked=True, input_is_relu=first_relu)
        _, T2, F2, C = y.shape
        lin = self.out[0] if self.use_pos_enc else self.out
        # the Linear is the only consumer of the ReLU output: its dgrad epilogue applies relu' (no mask pass over 15 M values)
        y = Fn.linear_fc_permuted(y.reshape(B * T2, F2 * C), lin.weight, lin.bias, C, F2, input_is_relu=True).view(B, T2, self.odim)
        if self.use_pos_enc:
            y = self.out[1](y)
        return y, self.out_lens(lens, T2, exact_lens)


# ------------------------------------------------------------------------------------------------
# Tacotron2 prenet / postnet (pre_postnets.py)
# ------------------------------------------------------------------------------------------------
class Prenet(nn.Module):
    """(Linear-ReLU-dropout)xN with dropout ALWAYS on (pre_postnets.py:53-66)."""

    def __init__(self, idim, n_layers=2, n_units=256, dropout_rate=0.5):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.prenet = nn.ModuleList()
        for layer in range(n_layers):
            self.prenet += [nn.Sequential(nn.Linear(idim if layer == 0 else n_units, n_units), nn.ReLU())]

    def forward(self, x):
        for blk in self.prenet:
            x = Fn.dropout(Fn.linear(x, blk[0].weight, blk[0].bias, act="relu"), self.dropout_rate)
        return x


class Postnet(nn.Module):
    """5 x (Conv1d k5 no-bias -> BatchNorm1d -> tanh (not last) -> dropout) (pre_postnets.py:69-185).
    Input/output here are channel-last (B, T, odim); padded frames are NOT masked."""

    def __init__(self, idim, odim, n_layers=5, n_chans=512, n_filts=5, dropout_rate=0.5, use_batch_norm=True):
        super().__init__()
        self.postnet = nn.ModuleList()
        for layer in range(n_layers):
            ichans = odim if layer == 0 else n_chans
            ochans = odim if layer == n_layers - 1 else n_chans
            mods = [nn.Conv1d(ichans, ochans, n_filts, stride=1, padding=(n_filts - 1) // 2, bias=False)]
            if use_batch_norm:
                mods.append(nn.BatchNorm1d(ochans))
            if layer != n_layers - 1:
                mods.append(nn.Tanh())
            mods.append(nn.Dropout(dropout_rate))
            self.postnet += [nn.Sequential(*mods)]
        self.dropout_rate, self.use_batch_norm = dropout_rate, use_batch_norm

    def forward(self, xs, lens=None):
        """lens: the Lens of xs.  In a captured step on a batch shorter than its padded shape (LensBank) its crop() marks the frames
        the reference's cropped tensor does not have: the convolutions read zero there, the BatchNorm statistics skip them."""
        n = len(self.postnet)
        p = self.dropout_rate if self.training else 0.0
        vl = crop_dev(lens)
        for i, blk in enumerate(self.postnet):
            act = "tanh" if i != n - 1 else None
            # a BatchNorm given vlens leaves zeros in the absent frames (and lets no gradient out of them): only the first
            # convolution -- or every one when there is no BatchNorm -- has to clear them itself
            xs = Fn.conv1d(xs, blk[0].weight, None, vlens=vl if (i == 0 or not self.use_batch_norm or not self.training) else None)
            if self.use_batch_norm:
                bn = blk[1]
                xs = Fn.batch_norm_act(xs, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                       self.training, act, p, bn.eps, bn.momentum, vlens=vl if self.training else None)
            else:
                xs = Fn.act_dropout(xs, act, p)
        return xs


# ------------------------------------------------------------------------------------------------
# Transformer encoder / decoder stacks (transformer/encoder.py, transformer/decoder.py)
# ------------------------------------------------------------------------------------------------
class MultiSequential(nn.Sequential):
    """Holder with the reference's container name; stacks are driven by run_stack()."""


class TransformerEncoder(nn.Module):
    def __init__(self, idim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                 positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 pos_enc_class=PositionalEncoding, normalize_before=True, concat_after=False,
                 positionwise_layer_type="linear", positionwise_conv_kernel_size=1, padding_idx=-1):
        super().__init__()
        if input_layer == "conv2d":
            self.embed = Conv2dSubsampling(idim, attention_dim, dropout_rate)
        elif input_layer == "conv2d-scaled-pos-enc":
            self.embed = Conv2dSubsampling(idim, attention_dim, dropout_rate, pos_enc_class(attention_dim, positional_dropout_rate))
        elif isinstance(input_layer, nn.Module):
            self.embed = nn.Sequential(input_layer, pos_enc_class(attention_dim, positional_dropout_rate))
        else:
            raise ValueError("unsupported input_layer: " + str(input_layer))
        self.normalize_before = normalize_before
        if positionwise_layer_type == "linear":
            ff = lambda: PositionwiseFeedForward(attention_dim, linear_units, dropout_rate)
        elif positionwise_layer_type == "conv1d":
            ff = lambda: MultiLayeredConv1d(attention_dim, linear_units, positionwise_conv_kernel_size, dropout_rate)
        else:
            raise NotImplementedError("Support only linear or conv1d.")
        self.encoders = MultiSequential(*[
            EncoderLayer(attention_dim, MultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate), ff(),
                         dropout_rate, normalize_before, concat_after) for _ in range(num_blocks)])
        if normalize_before:
            self.after_norm = LayerNorm(attention_dim)
        self.dropout_rate = dropout_rate

    def forward(self, xs, lens, exact_lens=False):
        if isinstance(self.embed, Conv2dSubsampling):
            xs, lens = self.embed(xs, lens, exact_lens)
        else:
            emb = self.embed[0]
            if isinstance(emb, nn.Embedding):
                xs = Fn.embedding(xs, emb.weight, emb.padding_idx)
            else:
                xs = emb(xs)
            xs = self.embed[1](xs)
        cut_name = getattr(self, "cut_name", None)      # data-parallel overlap (models' dp_plan): cut behind the input layer
        if cut_name is not None:
            xs = Fn.cut_point(xs, f"{cut_name}.0")
        xs = run_stack(self.encoders, xs, getattr(self, "after_norm", None), self.normalize_before, self.dropout_rate,
                       self.training, lens, cut_name=cut_name)
        return xs, lens


class Decoder(nn.Module):
    def __init__(self, odim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                 positional_dropout_rate=0.1, self_attention_dropout_rate=0.0, src_attention_dropout_rate=0.0,
                 input_layer="embed", use_output_layer=True, pos_enc_class=PositionalEncoding, normalize_before=True,
                 concat_after=False):
        super().__init__()
        if not isinstance(input_layer, nn.Module):
            raise NotImplementedError("only a torch.nn.Module input layer is supported")
        if use_output_layer:
            raise NotImplementedError("use_output_layer=True is not used by VTN / TTS")
        self.embed = nn.Sequential(input_layer, pos_enc_class(attention_dim, positional_dropout_rate))
        self.normalize_before = normalize_before
        self.decoders = MultiSequential(*[
            DecoderLayer(attention_dim, MultiHeadedAttention(attention_heads, attention_dim, self_attention_dropout_rate),
                         MultiHeadedAttention(attention_heads, attention_dim, src_attention_dropout_rate),
                         PositionwiseFeedForward(attention_dim, linear_units, dropout_rate), dropout_rate, normalize_before,
                         concat_after) for _ in range(num_blocks)])
        if normalize_before:
            self.after_norm = LayerNorm(attention_dim)
        self.output_layer = None
        self.dropout_rate = dropout_rate

    def embed_input(self, tgt):
        x = tgt
        for m in self.embed[0]:
            x = m(x) if isinstance(m, Prenet) else Fn.linear(x, m.weight, m.bias)
        return self.embed[1](x)

    def head(self, tgt, tgt_lens, causal=True):
        """Everything of the decoder that does not depend on the encoder: input layer, positional encoding and the
        self-attention block of the first layer.  In training the AR models run it on the auxiliary stream while the encoder
        runs (models/vtn.py); forward(..., head=...) continues from it."""
        x = self.embed_input(tgt)
        return self.decoders[0].self_block(x, tgt_lens, None, causal)

    def forward(self, tgt, tgt_lens, memory, mem_lens, causal=True, head=None):
        x = self.embed_input(tgt) if head is None else head[1]
        per_layer = None
        f = getattr(self, "_src_kv_all", None)       # stacked K/V weights of all source-attention blocks (optim.FlatAdam)
        if f is not None:
            # the memory is projected for ALL layers by one GEMM (N = layers * 2D); backward: one dgrad GEMM with
            # K = layers * 2D instead of one per layer plus the gradient adds over the memory's fan-out
            kv_all = Fn.linear(memory, f["w"], f["b"])
            per_layer = [{"kv": t} for t in Fn.split_cols(kv_all, len(self.decoders))]
        if head is not None:
            per_layer = per_layer if per_layer is not None else [{} for _ in self.decoders]
            per_layer[0] = dict(per_layer[0], head=head)
        x = run_stack(self.decoders, x, getattr(self, "after_norm", None), self.normalize_before, self.dropout_rate,
                      self.training, tgt_lens, memory, mem_lens, causal=causal, per_layer=per_layer)
        return x, tgt_lens
