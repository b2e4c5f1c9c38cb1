"""Transformer-TTS (text -> mel, autoregressive), MI355X-native drop-in for
`seq2seq_vc.models.TransformerTTS` (reference models/transformer_tts.py:14-326).  Shares every kernel
with VTN; only the encoder input layer (token Embedding + scaled positional encoding) differs."""
import torch
import torch.nn.functional as TF
from torch import nn

from .. import modules as Mo
from ..ops import functional as Fn
from ..ops import kernels as K
from .vtn import _ARSeq2Seq


class TransformerTTS(_ARSeq2Seq):
    def __init__(self, idim, odim, dprenet_layers=2, dprenet_units=256, adim=384, aheads=4, elayers=6, eunits=1536,
                 dlayers=6, dunits=1536, postnet_layers=5, postnet_filts=5, postnet_chans=256, dprenet_dropout_rate=0.5,
                 use_batch_norm=True, encoder_normalize_before=True, decoder_normalize_before=False,
                 encoder_concat_after=False, decoder_concat_after=False, decoder_reduction_factor=2, spk_embed_dim=None,
                 spk_embed_integration_type="add", initial_encoder_alpha=1.0, initial_decoder_alpha=1.0,
                 use_guided_attn_loss=False, num_heads_applied_guided_attn=2, num_layers_applied_guided_attn=2):
        nn.Module.__init__(self)
        self.idim, self.odim = idim, odim
        self.eos = idim - 1
        if spk_embed_dim is not None:
            raise NotImplementedError("speaker-embedding integration is out of scope (no recipe config uses it)")
        self.spk_embed_dim = None
        self.decoder_reduction_factor = decoder_reduction_factor
        self.use_guided_attn_loss = use_guided_attn_loss
        self.num_heads_applied_guided_attn = num_heads_applied_guided_attn
        self.num_layers_applied_guided_attn = num_layers_applied_guided_attn
        self.padding_idx = 0
        encoder_input_layer = nn.Embedding(num_embeddings=idim, embedding_dim=adim, padding_idx=self.padding_idx)
        self.encoder = Mo.TransformerEncoder(idim=idim, attention_dim=adim, attention_heads=aheads, linear_units=eunits,
                                             num_blocks=elayers, input_layer=encoder_input_layer,
                                             pos_enc_class=Mo.ScaledPositionalEncoding,
                                             normalize_before=encoder_normalize_before, concat_after=encoder_concat_after)
        self._build_decoder_side(idim, odim, dprenet_layers, dprenet_units, dprenet_dropout_rate, adim, aheads, dlayers, dunits,
                                 decoder_normalize_before, decoder_concat_after, decoder_reduction_factor, postnet_layers,
                                 postnet_chans, postnet_filts, use_batch_norm)
        self.encoder.embed[-1].alpha.data = torch.tensor(initial_encoder_alpha)
        self.decoder.embed[-1].alpha.data = torch.tensor(initial_decoder_alpha)

    def forward(self, xs, ilens, ys, labels, olens, spembs=None, *args, **kwargs):
        dev = ys.device
        il = Mo.Lens.of(ilens, dev)
        ol = Mo.Lens.of(olens, dev)
        if il.max() != xs.shape[1]:
            xs = xs[:, : il.max()]
        if ol.max() != ys.shape[1]:
            ys, labels = ys[:, : ol.max()], labels[:, : ol.max()]
        if xs.is_cuda and xs.dtype == torch.int64 and xs.stride(1) == 1:     # transformer_tts.py:139-142: append <eos>, one launch
            xs = K.append_eos(xs, il.dev, self.eos, self.padding_idx)
        else:
            xs = TF.pad(xs, [0, 1], "constant", self.padding_idx)
            xs = xs.scatter(1, il.dev.long().unsqueeze(1), self.eos)        # (no host round trip: capturable in a hipGraph)
        il1 = il.map(lambda v: v + 1)
        pre = self._decoder_head(ys, olens)
        hs, hs_lens = self.encoder(xs, il1)
        hs = Fn.cut_point(hs, "encoder_out")
        after, before, logits, ys_, labels_, olens_, olens_in = self._teacher_forced(hs, hs_lens, ys, labels, olens, pre=pre)
        att_ws = []
        if self.use_guided_attn_loss:
            n = len(self.decoder.decoders)
            for idx, li in enumerate(reversed(range(n))):
                att_ws.append(self.decoder.decoders[li].src_attn.attn[:, : self.num_heads_applied_guided_attn])
                if idx + 1 == self.num_layers_applied_guided_attn:
                    break
            att_ws = torch.cat(att_ws, dim=1)
        ilens_out = Mo.tag_lens((ilens + 1) if isinstance(ilens, torch.Tensor) else torch.tensor(il1.host), il1)
        return after, before, logits, ys_, labels_, olens_, (att_ws, ilens_out, olens_in)

    @torch.no_grad()
    def inference(self, x, inference_args, spemb=None, *args, **kwargs):
        x = TF.pad(x, [0, 1], "constant", self.eos).unsqueeze(0)
        hs, _ = self.encoder(x, None)
        return self._decode_loop(hs, inference_args["threshold"], inference_args["minlenratio"], inference_args["maxlenratio"])
