"""VTN -- Voice Transformer Network (mel -> mel, autoregressive), MI355X-native.

Drop-in for `seq2seq_vc.models.VTN` (reference models/vtn.py): same constructor keywords
(:15-62), same `forward` inputs / 7-tuple outputs (:207-300), same `inference` (:302-394), same
state_dict keys.  All arithmetic runs in the HIP kernels behind seq2seq_vc_amd.ops.functional.
"""
import logging

import torch
from torch import nn

from .. import modules as Mo
from ..ops import functional as Fn
from ..ops import kernels as K




class _ARSeq2Seq(nn.Module):
    """Shared decoder-side logic of VTN and TransformerTTS (vtn.py:227-300 == transformer_tts.py:160-229)."""

    def _build_decoder_side(self, idim, odim, dprenet_layers, dprenet_units, dprenet_dropout_rate, adim, aheads, dlayers,
                            dunits, decoder_normalize_before, decoder_concat_after, decoder_reduction_factor, postnet_layers,
                            postnet_chans, postnet_filts, use_batch_norm):
        decoder_input_layer = nn.Sequential(
            Mo.Prenet(idim=odim, n_layers=dprenet_layers, n_units=dprenet_units, dropout_rate=dprenet_dropout_rate),
            nn.Linear(dprenet_units, adim))
        self.decoder = Mo.Decoder(odim=-1, attention_dim=adim, attention_heads=aheads, linear_units=dunits, num_blocks=dlayers,
                                  input_layer=decoder_input_layer, use_output_layer=False,
                                  pos_enc_class=Mo.ScaledPositionalEncoding, normalize_before=decoder_normalize_before,
                                  concat_after=decoder_concat_after)
        self.feat_out = nn.Linear(adim, odim * decoder_reduction_factor)
        self.prob_out = nn.Linear(adim, decoder_reduction_factor)
        self.postnet = Mo.Postnet(idim=idim, odim=odim, n_layers=postnet_layers, n_chans=postnet_chans, n_filts=postnet_filts,
                                  use_batch_norm=use_batch_norm)

    def dp_plan(self):
        """Stages of the data-parallel backward pass (distributed.OverlappedBackward): the decoder side finishes first and its
        gradients (half of the parameters) travel while the encoder's backward pass -- below the cut at the encoder output --
        runs; the encoder is cut once more behind its input layer, so that the bucket left over when the backward pass ends
        -- the only exchange nothing hides -- is the input layer's alone (VTN vc1: 62.8 | 42.6 | 16.5 MB instead of 62.8 | 59.1).
        A cut costs ~0.09 ms of the step (graph boundary + lost overlap; `bench.py --force-dist --stage-times`: stage graphs
        2.83 + 0.32 + 0.32 + 0.59 ms with the layer stack cut in the middle as well -- round 2's plan -- against 3.78 ms for the
        uncut backward pass), more than the middle cut saves: the 42.6 MB of the layer stack travel behind the input layer's
        0.6 ms from 75 GB/s of all-reduce bandwidth on.  One loss key: "loss"."""
        dec_side = [m for m in (self.decoder, self.feat_out, self.prob_out, self.postnet) if m is not None]
        enc = self.encoder
        layers = list(enc.encoders)
        tail = [m for m in (getattr(enc, "after_norm", None),) if m is not None]
        embed = [m for n, m in enc.named_children() if n not in ("encoders", "after_norm")]
        # Round 4: the decoder's head (input layer + positional encoding + the first layer's self-attention block: what runs on the
        # auxiliary stream beside the encoder) is cut off the decoder's stage: its backward pass runs in the ENCODER's stage, on the
        # auxiliary stream beside the layer stack's, as it does in the one-graph step; its parameters travel with that bucket.
        d0 = self.decoder.decoders[0]
        hm = [self.decoder.embed, d0.self_attn, d0.norm1] + ([d0.norm2] if d0.normalize_before else [])
        hp = {id(p) for m in hm for p in m.parameters()}
        dec_side = [p for m in dec_side for p in m.parameters() if id(p) not in hp]
        if len(layers) < 2 or not embed:
            return [{"root": "loss:loss", "modules": dec_side},
                    {"root": "cut:encoder_out", "branch_root": "cut:decoder_head", "modules": [enc] + hm}]
        enc.cut_name = "encoder"            # names the cut points inside Encoder.forward / run_stack
        return [{"root": "loss:loss", "modules": dec_side},
                {"root": "cut:encoder_out", "branch_root": "cut:decoder_head", "modules": layers + tail + hm},
                {"root": "cut:encoder.0", "modules": embed}]

    def _decoder_head(self, ys, olens, labels=None):
        """Teacher-forcing inputs of the decoder and -- in training, on the GPU -- the part of the decoder that does not see the
        encoder (input layer + positional encoding + the first layer's self-attention block), started on the auxiliary stream
        so that it runs beside the encoder (forward) and beside the encoder's backward pass (autograd runs a node on the stream
        of its forward op).  The shifted decoder input and the stop labels of the trimmed targets (`labels`, a few tiny torch
        kernels that depend on the batch only) go there with it instead of sitting on the main chain.
        -> state for _teacher_forced(..., pre=state)."""
        r = self.decoder_reduction_factor
        dev = ys.device
        olens_h = Mo.Lens.of(olens, dev)
        if r > 1:
            ys_in = ys[:, r - 1::r]
            olens_in_h = olens_h.map(lambda v: v // r)
        else:
            ys_in, olens_in_h = ys, olens_h
        def shifted():
            if ys.is_cuda and ys.dtype == torch.float32 and ys.stride(2) == 1 and ys.stride(1) == ys.shape[2]:
                return K.decoder_input(ys, r, Fn.compute_dtype())        # shift + stride r + cast, one launch (csrc/glue.hip)
            return torch.cat([ys_in.new_zeros((ys_in.shape[0], 1, ys_in.shape[2])), ys_in[:, :-1]], dim=1)

        head, stop = None, None
        if self.training and ys.is_cuda and torch.is_grad_enabled():
            def run():
                y0 = shifted()
                lab = self._stop_labels(labels, olens_h)[1] if labels is not None else None
                return y0, lab, self.decoder.head(Fn.to_compute(y0), olens_in_h, causal=True)

            ys_in, stop, head = Fn.branch_run(run, uses=(ys, labels, olens_in_h.dev, olens_h.dev))
        else:
            ys_in = shifted()
        return olens_h, olens_in_h, ys_in, head, stop

    def _stop_labels(self, labels, olens_h):
        """Targets trimmed to a multiple of the reduction factor: (trimmed lengths, labels with the stop flag at the last kept
        frame) -- reference models/vtn.py:253-260."""
        r = self.decoder_reduction_factor
        olens_out_h = olens_h.map(lambda v: v - v % r)
        mx = olens_out_h.max()
        if labels.is_cuda and labels.dtype == torch.float32 and labels.stride(1) == 1:
            return olens_out_h, K.stop_labels(labels, olens_out_h.dev, mx)
        idx = (olens_out_h.dev.long() - 1).unsqueeze(1)
        return olens_out_h, torch.scatter(labels[:, :mx], 1, idx, 1.0)

    def _teacher_forced(self, hs, hs_lens, ys, labels, olens, pre=None):
        r, odim = self.decoder_reduction_factor, self.odim
        olens_h, olens_in_h, ys_in, head, stop = pre if pre is not None else self._decoder_head(ys, olens)
        if head is not None:
            Fn.branch_join(*head, ys_in, stop)
            # gradient cut for the staged (data-parallel) backward pass: the head's backward pass -- the last thing of the decoder's,
            # on the auxiliary stream -- overlaps the encoder's in the one-graph step; a stage plan may run it in the encoder's stage
            # (dp_plan).  Identity outside distributed.OverlappedBackward.
            head = Fn.cut_point(tuple(head), "decoder_head")
            zs, _ = self.decoder(None, olens_in_h, hs, hs_lens, causal=True, head=head)
        else:
            zs, _ = self.decoder(Fn.to_compute(ys_in), olens_in_h, hs, hs_lens, causal=True)
        # zs feeds feat_out AND prob_out: prob_out takes it from feat_out's pass-through alias, so its gradient rides in feat_out's
        # data-gradient GEMM instead of in an element-wise add of autograd's accumulation
        if self.training and torch.is_grad_enabled() and zs.requires_grad:
            before, zs = Fn.linear(zs, self.feat_out.weight, self.feat_out.bias, passthrough=True)
            before = before.view(zs.size(0), -1, odim)
        else:
            before = Fn.linear(zs, self.feat_out.weight, self.feat_out.bias).view(zs.size(0), -1, odim)
        b_post = b_res = before
        if self.postnet is not None:        # three consumers (loss, residual, Postnet): their gradients meet in one launch
            before, b_res, b_post = Fn.fan_out(before, 3)
        # captured step (modules.LensBank): `before` has olens_in * r frames per utterance; the Postnet must not see the frames the
        # reference's cropped batch does not have (vtn.py:208-214: ys is cropped to the longest utterance before anything runs)
        post_lens = None
        if olens_in_h.cap is not None and self.postnet is not None:
            post_lens = olens_in_h if r == 1 else olens_in_h.map(lambda v: v * r)
        if self.training and zs.is_cuda and torch.is_grad_enabled() and self.postnet is not None:
            # the stop-token projection (384 -> r columns: a GEMM with one output column forward, a rank-1 product backward, both on
            # slow general kernels) beside the Postnet instead of in front of it: the auxiliary stream is idle here, and autograd runs
            # its backward node there too -- beside the Postnet's backward pass instead of between it and feat_out's
            (logits,) = Fn.branch_run(lambda: (Fn.linear(zs, self.prob_out.weight, self.prob_out.bias).view(zs.size(0), -1),), uses=(zs,))
            after = Fn.add_dropout(b_res, self.postnet(b_post, post_lens), 0.0)
            Fn.branch_join(logits)
        else:
            logits = Fn.linear(zs, self.prob_out.weight, self.prob_out.bias).view(zs.size(0), -1)
            after = Fn.add_dropout(b_res, self.postnet(b_post, post_lens), 0.0) if self.postnet is not None else before
        olens_out = olens
        if r > 1:
            if min(olens_h.host) < r:
                raise AssertionError("Output length must be greater than or equal to reduction factor.")
            if stop is not None:                 # made beside the encoder (_decoder_head)
                olens_out_h = olens_h.map(lambda v: v - v % r)
                labels = stop
            else:
                olens_out_h, labels = self._stop_labels(labels, olens_h)
            new = list(olens_out_h.host)
            olens_out = Mo.tag_lens(olens.new_tensor(new) if isinstance(olens, torch.Tensor) else torch.tensor(new), olens_out_h)
            ys = ys[:, :olens_out_h.max()]
        olens_in = Mo.tag_lens(olens.new_tensor(olens_in_h.host) if isinstance(olens, torch.Tensor) else torch.tensor(olens_in_h.host),
                               olens_in_h)
        return after, before, logits, ys, labels, olens_out, olens_in

    def _decode_loop(self, hs, threshold, minlenratio, maxlenratio):
        """Step-wise generation for one utterance: static K/V cache + captured step graph (decode.py)."""
        from ..decode import decode
        args = {"threshold": threshold, "minlenratio": minlenratio, "maxlenratio": maxlenratio}
        return decode(self, hs, [hs.size(1)], args)[0]

    def _decode_loop_recompute(self, hs, threshold, minlenratio, maxlenratio):
        """The reference's schedule, kept as a cross-check of the cached path: every step re-evaluates the
        decoder over the whole prefix (the reference's per-layer cache changes cost, not values)."""
        r, odim = self.decoder_reduction_factor, self.odim
        maxlen = int(hs.size(1) * maxlenratio / r)
        minlen = int(hs.size(1) * minlenratio / r)
        ys = torch.zeros((1, 1, odim), dtype=hs.dtype, device=hs.device)
        outs, probs, atts = [], [], []
        idx = 0
        while True:
            idx += 1
            zs, _ = self.decoder(ys, None, hs, None, causal=True)
            z = zs[:, -1]
            outs.append(Fn.linear(z, self.feat_out.weight, self.feat_out.bias).view(r, odim))
            probs.append(torch.sigmoid(Fn.linear(z, self.prob_out.weight, self.prob_out.bias).float())[0])
            ys = torch.cat((ys, outs[-1][-1].view(1, 1, odim)), dim=1)
            atts.append(torch.stack([d.src_attn.attn[0, :, -1] for d in self.decoder.decoders]))  # (layers, H, T)
            if int((probs[-1] >= threshold).sum()) > 0 or idx >= maxlen:
                if idx < minlen:
                    continue
                o = torch.cat(outs, dim=0).unsqueeze(0)             # (1, L, odim) channel-last
                if self.postnet is not None:
                    o = Fn.add_dropout(o, self.postnet(o.contiguous()), 0.0)
                return o.squeeze(0).float(), torch.cat(probs, dim=0), torch.stack(atts, dim=2).float()


class VTN(_ARSeq2Seq):
    def __init__(self, idim, odim, dprenet_layers=2, dprenet_units=256, adim=384, aheads=4, encoder_type="transformer",
                 decoder_type="transformer", elayers=6, eunits=1536, dlayers=6, dunits=1536, postnet_layers=5,
                 postnet_filts=5, postnet_chans=256, positionwise_layer_type: str = "linear",
                 positionwise_conv_kernel_size: int = 1, dprenet_dropout_rate=0.5,
                 transformer_enc_dropout_rate: float = 0.1, transformer_enc_positional_dropout_rate: float = 0.1,
                 transformer_enc_attn_dropout_rate: float = 0.1, use_batch_norm=True, encoder_normalize_before=True,
                 decoder_normalize_before=False, encoder_concat_after=False, decoder_concat_after=False,
                 decoder_reduction_factor=2, spk_embed_dim=None, spk_embed_integration_type="add",
                 initial_encoder_alpha=1.0, initial_decoder_alpha=1.0, use_guided_attn_loss=False,
                 num_heads_applied_guided_attn=2, num_layers_applied_guided_attn=2, conformer_rel_pos_type: str = "legacy",
                 conformer_pos_enc_layer_type: str = "rel_pos", conformer_self_attn_layer_type: str = "rel_selfattn",
                 use_macaron_style_in_conformer: bool = True, use_cnn_in_conformer: bool = True, zero_triu: bool = False,
                 conformer_enc_kernel_size: int = 7, conformer_dec_kernel_size: int = 31):
        nn.Module.__init__(self)
        self.idim, self.odim = idim, odim
        self.spk_embed_dim = spk_embed_dim
        if spk_embed_dim is not None:
            raise NotImplementedError("speaker-embedding integration is out of scope (no recipe config uses it)")
        self.decoder_reduction_factor = decoder_reduction_factor
        self.use_guided_attn_loss = use_guided_attn_loss
        self.num_heads_applied_guided_attn = num_heads_applied_guided_attn
        self.num_layers_applied_guided_attn = num_layers_applied_guided_attn
        self.encoder_type, self.decoder_type = encoder_type, decoder_type

        if encoder_type == "conformer":  # vtn.py:83-104 compatibility fallback
            if conformer_rel_pos_type == "legacy":
                if conformer_pos_enc_layer_type == "rel_pos":
                    conformer_pos_enc_layer_type = "legacy_rel_pos"
                    logging.warning("Fallback to conformer_pos_enc_layer_type = 'legacy_rel_pos' due to the compatibility.")
                if conformer_self_attn_layer_type == "rel_selfattn":
                    conformer_self_attn_layer_type = "legacy_rel_selfattn"
                    logging.warning("Fallback to conformer_self_attn_layer_type = 'legacy_rel_selfattn' due to the compatibility.")
            elif conformer_rel_pos_type == "latest":
                assert conformer_pos_enc_layer_type != "legacy_rel_pos"
                assert conformer_self_attn_layer_type != "legacy_rel_selfattn"
            else:
                raise ValueError(f"Unknown rel_pos_type: {conformer_rel_pos_type}")

        if encoder_type == "transformer":
            self.encoder = Mo.TransformerEncoder(
                idim=idim, attention_dim=adim, attention_heads=aheads, linear_units=eunits, num_blocks=elayers,
                input_layer="conv2d-scaled-pos-enc", pos_enc_class=Mo.ScaledPositionalEncoding,
                normalize_before=encoder_normalize_before, concat_after=encoder_concat_after,
                positionwise_layer_type=positionwise_layer_type,
                positionwise_conv_kernel_size=positionwise_conv_kernel_size, dropout_rate=transformer_enc_dropout_rate)
        elif encoder_type == "conformer":
            from ..conformer import ConformerEncoder
            self.encoder = ConformerEncoder(
                idim=idim, attention_dim=adim, attention_heads=aheads, linear_units=eunits, num_blocks=elayers,
                input_layer="conv2d", normalize_before=encoder_normalize_before, concat_after=encoder_concat_after,
                positionwise_layer_type=positionwise_layer_type,
                positionwise_conv_kernel_size=positionwise_conv_kernel_size, dropout_rate=transformer_enc_dropout_rate,
                positional_dropout_rate=transformer_enc_positional_dropout_rate,
                attention_dropout_rate=transformer_enc_attn_dropout_rate, macaron_style=use_macaron_style_in_conformer,
                pos_enc_layer_type=conformer_pos_enc_layer_type, selfattention_layer_type=conformer_self_attn_layer_type,
                use_cnn_module=use_cnn_in_conformer, cnn_module_kernel=conformer_enc_kernel_size, zero_triu=zero_triu)
        else:
            raise NotImplementedError

        self._build_decoder_side(idim, odim, dprenet_layers, dprenet_units, dprenet_dropout_rate, adim, aheads, dlayers, dunits,
                                 decoder_normalize_before, decoder_concat_after, decoder_reduction_factor, postnet_layers,
                                 postnet_chans, postnet_filts, use_batch_norm)
        self._reset_parameters(initial_encoder_alpha, initial_decoder_alpha)

    def _reset_parameters(self, init_enc_alpha: float, init_dec_alpha: float):
        if self.encoder_type == "transformer":
            self.encoder.embed[-1].alpha.data = torch.tensor(init_enc_alpha)
        if self.decoder_type == "transformer":
            self.decoder.embed[-1].alpha.data = torch.tensor(init_dec_alpha)

    def forward(self, xs, ilens, ys, labels, olens, spembs=None, *args, **kwargs):
        """xs (B,Tmax,idim), ilens (B,), ys (B,Lmax,odim), labels (B,Lmax), olens (B,) ->
        (after_outs, before_outs, logits, ys, labels, olens, (att_ws, ilens_ds_st, olens_in)).
        `ilens`/`olens` are best passed as CPU LongTensors (as the collater yields them): lengths are
        consumed on the host to size kernels, never to build mask tensors."""
        dev = xs.device
        K.reset_op_counter() if kwargs.get("_reset_seed_counter", False) else None
        il = Mo.Lens.of(ilens, dev)
        ol = Mo.Lens.of(olens, dev)
        if il.max() != xs.shape[1]:
            xs = xs[:, : il.max()]
        if ol.max() != ys.shape[1]:
            ys, labels = ys[:, : ol.max()], labels[:, : ol.max()]
        pre = self._decoder_head(ys, olens, labels if self.decoder_reduction_factor > 1 else None)
        hs, hs_lens = self.encoder(Fn.to_compute(xs), il)
        hs = Fn.cut_point(hs, "encoder_out")      # data-parallel overlap: decoder-side gradients travel during the encoder's backward
        after, before, logits, ys_, labels_, olens_, olens_in = self._teacher_forced(hs, hs_lens, ys, labels, olens, pre=pre)
        il_ds = il.map(lambda v: ((v - 2 + 1) // 2 - 2 + 1) // 2)
        ilens_ds_st = Mo.tag_lens(torch.tensor(list(il_ds.host), dtype=ilens.dtype if isinstance(ilens, torch.Tensor) else torch.long,
                                               device=ilens.device if isinstance(ilens, torch.Tensor) else "cpu"), il_ds)
        att_ws = [self.decoder.decoders[i].src_attn.attn for i in reversed(range(len(self.decoder.decoders)))]
        return after, before, logits, ys_, labels_, olens_, (att_ws, ilens_ds_st, olens_in)

    @torch.no_grad()
    def inference(self, x, inference_args, spemb=None, *args, **kwargs):
        """x (T, idim) -> (outs (L, odim), probs (L,), att_ws (#layers, #heads, L/r, T_enc))."""
        hs, _ = self.encoder(Fn.to_compute(x.unsqueeze(0)), None)
        return self._decode_loop(hs, inference_args["threshold"], inference_args["minlenratio"], inference_args["maxlenratio"])

    @torch.no_grad()
    def inference_batch(self, xs, ilens, inference_args, poll=16):
        """Several utterances in lockstep: xs (B, Tmax, idim) zero-padded, ilens (B,).  Returns one
        `inference` result per utterance.  (Extension: the reference decodes utterance by utterance,
        bin/vc_decode.py:264-306.  Each row is given the encoder length it has when processed alone, so
        that padding never enters an utterance and the frames equal those of single-utterance decoding.)"""
        from ..decode import decode
        lens = Mo.Lens.of(ilens, xs.device)
        hs, hlens = self.encoder(Fn.to_compute(xs), lens, exact_lens=True)
        return decode(self, hs, list(hlens.host), inference_args, poll=poll)
