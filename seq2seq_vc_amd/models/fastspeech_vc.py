"""FastSpeech2-style non-autoregressive VC with teacher durations, MI355X-native drop-in for
`seq2seq_vc.models.FastSpeechVC` (reference models/fastspeech_vc.py:21-513; recipe egs/arctic/vc2/conf/fs2_vc.melmelmel.v1.yaml).

Same constructor keywords, `forward` / `inference` signatures and return tuples, state_dict keys.  It reuses every Conformer /
Transformer kernel of the other models; what it adds is the LengthRegulator (repeat-interleave expansion, csrc/lenreg.hip)
driven by the teacher's durations, the deterministic DurationPredictor on a separately projected input, and the
DurationCalculator that extracts those durations from a teacher's attention maps (utils.DurationCalculator).

Reference facts reproduced: only the Conformer decoder is constructible (the Transformer decoder branch names an undefined
`pos_enc_class`, fastspeech_vc.py:180); with `encoder_input_layer == "conv2d"` (and with the Transformer encoder, whose input
layer is always the Conv2d front-end) the encoder subsamples by 4 and `ilens` is adjusted only in the Conformer/conv2d case
(:279-280); durations arrive as a LongTensor (B, Tmax) and are multiplied by `teacher_model_decoder_reduction_factor`."""
import logging

import torch
from torch import nn

from .. import modules as Mo
from ..conformer import ConformerEncoder
from ..ops import functional as Fn
from ..ops import functional_aas as FA
from ..sdp import DurationPredictor


class LengthRegulator(nn.Module):
    """modules/length_regulator.py:46-97 on the GPU: xs (B,Tmax,D), ds (B,Tmax) integer durations -> (B, max total, D)."""

    def __init__(self, pad_value=0.0):
        super().__init__()
        self.pad_value = pad_value

    def forward(self, xs, ds, alpha=1.0):
        """`ds` is best a CPU LongTensor (as the collater yields it): the output length is the largest per-utterance
        total, which the host has to know to size the output -- a device tensor costs one synchronising copy here, exactly
        where the reference's pad_list takes its `max(x.size(0))`."""
        ds_host = ds.detach().to("cpu")
        if alpha != 1.0:
            assert alpha > 0
            ds_host = torch.round(ds_host.float() * alpha).long()
        ds_host = ds_host.long()
        if int(ds_host.sum()) == 0:
            logging.warning("predicted durations includes all 0 sequences. fill the first element with 1.")
            ds_host = ds_host.clone()
            ds_host[ds_host.sum(dim=1).eq(0)] = 1
        Tout = int(ds_host.sum(dim=1).max())
        ds_dev = ds_host.to(device=xs.device, dtype=torch.int32)
        return FA.length_regulate(xs, ds_dev.contiguous(), Tout, self.pad_value)


class FastSpeechVC(nn.Module):
    def __init__(self, idim, odim, adim: int = 384, aheads: int = 4, elayers: int = 6, eunits: int = 1536, dlayers: int = 6,
                 dunits: int = 1536, postnet_layers: int = 5, postnet_chans: int = 512, postnet_filts: int = 5,
                 positionwise_layer_type: str = "conv1d", positionwise_conv_kernel_size: int = 1, use_scaled_pos_enc: bool = True,
                 use_batch_norm: bool = True, encoder_input_layer: str = "linear", encoder_input_conv_kernel_size: int = 3,
                 encoder_normalize_before: bool = False, decoder_normalize_before: bool = False,
                 encoder_concat_after: bool = False, decoder_concat_after: bool = False,
                 duration_predictor_use_encoder_outputs: bool = True, duration_predictor_input_dim: int = None,
                 duration_predictor_layers: int = 2, duration_predictor_chans: int = 384, duration_predictor_kernel_size: int = 3,
                 encoder_reduction_factor: int = 1, decoder_reduction_factor: int = 1, encoder_type: str = "transformer",
                 decoder_type: str = "transformer", conformer_pos_enc_layer_type: str = "rel_pos",
                 conformer_self_attn_layer_type: str = "rel_selfattn", use_macaron_style_in_conformer: bool = True,
                 use_cnn_in_conformer: bool = True, conformer_enc_kernel_size: int = 7, conformer_dec_kernel_size: int = 31,
                 spk_embed_dim: int = None, spk_embed_integration_type: str = "add", transformer_enc_dropout_rate: float = 0.1,
                 transformer_enc_positional_dropout_rate: float = 0.1, transformer_enc_attn_dropout_rate: float = 0.1,
                 transformer_dec_dropout_rate: float = 0.1, transformer_dec_positional_dropout_rate: float = 0.1,
                 transformer_dec_attn_dropout_rate: float = 0.1, duration_predictor_dropout_rate: float = 0.1,
                 postnet_dropout_rate: float = 0.5, init_type: str = "xavier_uniform", init_enc_alpha: float = 1.0,
                 init_dec_alpha: float = 1.0, use_masking: bool = False, use_weighted_masking: bool = False,
                 teacher_model_decoder_reduction_factor: int = 4):
        nn.Module.__init__(self)
        self.idim, self.odim = idim, odim
        if spk_embed_dim is not None:
            raise NotImplementedError("speaker-embedding integration is out of scope (no recipe config uses it)")
        self.spk_embed_dim = None
        self.encoder_reduction_factor, self.decoder_reduction_factor = encoder_reduction_factor, decoder_reduction_factor
        self.encoder_type, self.decoder_type = encoder_type, decoder_type
        self.use_scaled_pos_enc, self.encoder_input_layer = use_scaled_pos_enc, encoder_input_layer
        self.teacher_model_decoder_reduction_factor = teacher_model_decoder_reduction_factor
        self.duration_predictor_use_encoder_outputs = duration_predictor_use_encoder_outputs
        if encoder_type == "transformer":
            self.encoder = Mo.TransformerEncoder(
                idim=idim, attention_dim=adim, attention_heads=aheads, linear_units=eunits, num_blocks=elayers,
                input_layer="conv2d-scaled-pos-enc", pos_enc_class=Mo.ScaledPositionalEncoding,
                normalize_before=encoder_normalize_before, concat_after=encoder_concat_after,
                positionwise_layer_type=positionwise_layer_type, positionwise_conv_kernel_size=positionwise_conv_kernel_size,
                dropout_rate=transformer_enc_dropout_rate)
        elif encoder_type == "conformer":
            self.encoder = ConformerEncoder(
                idim=idim * encoder_reduction_factor, attention_dim=adim, attention_heads=aheads, linear_units=eunits,
                num_blocks=elayers, input_layer=encoder_input_layer, dropout_rate=transformer_enc_dropout_rate,
                positional_dropout_rate=transformer_enc_positional_dropout_rate,
                attention_dropout_rate=transformer_enc_attn_dropout_rate, normalize_before=encoder_normalize_before,
                concat_after=encoder_concat_after, positionwise_layer_type=positionwise_layer_type,
                positionwise_conv_kernel_size=positionwise_conv_kernel_size, macaron_style=use_macaron_style_in_conformer,
                pos_enc_layer_type=conformer_pos_enc_layer_type, selfattention_layer_type=conformer_self_attn_layer_type,
                use_cnn_module=use_cnn_in_conformer, cnn_module_kernel=conformer_enc_kernel_size)
        else:
            raise NotImplementedError
        self.duration_predictor = DurationPredictor(idim=adim, n_layers=duration_predictor_layers, n_chans=duration_predictor_chans,
                                                    kernel_size=duration_predictor_kernel_size,
                                                    dropout_rate=duration_predictor_dropout_rate)
        if not self.duration_predictor_use_encoder_outputs:
            self.duration_predictor_projection = Mo.Conv2dSubsampling(duration_predictor_input_dim, adim, 0.0, use_pos_enc=False)
        self.length_regulator = LengthRegulator()
        if decoder_type == "conformer":
            self.decoder = ConformerEncoder(
                idim=0, attention_dim=adim, attention_heads=aheads, linear_units=dunits, num_blocks=dlayers, input_layer=None,
                dropout_rate=transformer_dec_dropout_rate, positional_dropout_rate=transformer_dec_positional_dropout_rate,
                attention_dropout_rate=transformer_dec_attn_dropout_rate, normalize_before=decoder_normalize_before,
                concat_after=decoder_concat_after, positionwise_layer_type=positionwise_layer_type,
                positionwise_conv_kernel_size=positionwise_conv_kernel_size, macaron_style=use_macaron_style_in_conformer,
                pos_enc_layer_type=conformer_pos_enc_layer_type, selfattention_layer_type=conformer_self_attn_layer_type,
                use_cnn_module=use_cnn_in_conformer, cnn_module_kernel=conformer_dec_kernel_size)
        elif decoder_type == "transformer":
            raise NotImplementedError("the reference cannot construct this branch either (undefined pos_enc_class, "
                                      "models/fastspeech_vc.py:180); use decoder_type='conformer'")
        else:
            raise ValueError(f"{decoder_type} is not supported.")
        self.feat_out = nn.Linear(adim, odim * decoder_reduction_factor)
        self.postnet = None if postnet_layers == 0 else Mo.Postnet(
            idim=idim, odim=odim, n_layers=postnet_layers, n_chans=postnet_chans, n_filts=postnet_filts,
            use_batch_norm=use_batch_norm, dropout_rate=postnet_dropout_rate)
        if self.encoder_type == "transformer":
            self.encoder.embed[-1].alpha.data = torch.tensor(init_enc_alpha)
        self.decoder.cut_name = "decoder"

    def dp_plan(self):
        """Data-parallel backward stages: decoder side (postnet, feat_out, decoder), then -- below the cut at the encoder
        output -- the encoder; the duration predictor's loss is its own root.  Loss keys: "decoder", "duration"."""
        dec_side = [m for m in (self.decoder, self.feat_out, self.postnet) if m is not None]
        side = [self.duration_predictor] + ([self.duration_predictor_projection] if hasattr(self, "duration_predictor_projection") else [])
        return [{"root": "loss:decoder", "modules": dec_side}, {"root": "loss:duration", "modules": side},
                {"root": "cut:encoder_out", "modules": [self.encoder]}]

    def _forward(self, xs, ilens, olens=None, ds=None, dp_inputs=None, dplens=None, spembs=None, is_inference=False, alpha=1.0):
        dev = xs.device
        il = Mo.Lens.of(ilens, dev)
        er = self.encoder_reduction_factor
        if er > 1:
            b, tmax, dim = xs.shape
            if tmax % er != 0:
                xs = xs[:, : -(tmax % er)]
            xs = xs.contiguous().view(b, tmax // er, dim * er)
            il = il.map(lambda v: v // er)
        hs, _ = self.encoder(Fn.to_compute(xs), il)
        hs = Fn.cut_point(hs, "encoder_out")
        if self.encoder_input_layer == "conv2d":        # (:279-280; the Transformer encoder's front-end does not adjust ilens)
            il = il.map(lambda v: ((v - 2 + 1) // 2 - 2 + 1) // 2)
        if self.duration_predictor_use_encoder_outputs:
            dpi = hs
        else:
            dpi, _ = self.duration_predictor_projection(Fn.to_compute(dp_inputs), None)
            dpi = FA.interp_nearest(dpi, hs.shape[1])
        if is_inference:
            d_outs = self.duration_predictor.inference(dpi)
            hs = self.length_regulator(hs, d_outs * self.teacher_model_decoder_reduction_factor, alpha)
        else:
            d_outs = self.duration_predictor(dpi, il.clamp(dpi.shape[1]))
            hs = self.length_regulator(hs, ds * self.teacher_model_decoder_reduction_factor)
        dec_lens = None
        if olens is not None and not is_inference:
            ol = Mo.Lens.of(olens, dev)
            dec_lens = ol.map(lambda v: v // self.decoder_reduction_factor) if self.decoder_reduction_factor > 1 else ol
            dec_lens = dec_lens.clamp(hs.shape[1])
        zs, _ = self.decoder(hs, dec_lens)
        before = Fn.linear(zs, self.feat_out.weight, self.feat_out.bias).view(zs.size(0), -1, self.odim)
        after = before if self.postnet is None else Fn.add_dropout(before, self.postnet(before), 0.0)
        ilens_out = ilens.new_tensor(list(il.host)) if isinstance(ilens, torch.Tensor) else torch.tensor(list(il.host))
        return before, after, d_outs, ilens_out

    def forward(self, src_speech, src_speech_lengths, tgt_speech, tgt_speech_lengths, durations, durations_lengths, dp_inputs=None,
                dp_lengths=None, spembs=None):
        """-> (before_outs, after_outs, d_outs, ilens, olens, ys)   (models/fastspeech_vc.py:343-407)."""
        dev = src_speech.device
        il, ol = Mo.Lens.of(src_speech_lengths, dev), Mo.Lens.of(tgt_speech_lengths, dev)
        xs, ys = src_speech[:, : il.max()], tgt_speech[:, : ol.max()]
        ds = durations[:, : int(torch.as_tensor(durations_lengths).max())]
        before, after, d_outs, ilens_ = self._forward(xs, src_speech_lengths, tgt_speech_lengths, ds, dp_inputs=dp_inputs,
                                                      dplens=dp_lengths, spembs=spembs, is_inference=False)
        olens = tgt_speech_lengths
        if self.decoder_reduction_factor > 1:
            new = [v - v % self.decoder_reduction_factor for v in ol.host]
            olens = tgt_speech_lengths.new_tensor(new) if isinstance(tgt_speech_lengths, torch.Tensor) else torch.tensor(new)
            ys = ys[:, : max(new)]
        return before, after, d_outs, ilens_, olens, ys

    @torch.no_grad()
    def inference(self, src_speech, tgt_speech=None, spembs=None, durations=None, dp_input=None, alpha=1.0, use_teacher_forcing=False):
        x = src_speech
        ilens = torch.tensor([x.shape[0]], dtype=torch.long)
        dpi = None if dp_input is None else dp_input.unsqueeze(0)
        if use_teacher_forcing:
            _, outs, d_outs, _ = self._forward(x.unsqueeze(0), ilens, ds=durations.unsqueeze(0), dp_inputs=dpi)
        else:
            _, outs, d_outs, _ = self._forward(x.unsqueeze(0), ilens, dp_inputs=dpi, is_inference=True, alpha=alpha)
        return outs[0].float(), d_outs[0]
