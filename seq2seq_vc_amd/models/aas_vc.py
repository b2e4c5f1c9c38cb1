"""AAS-VC (non-autoregressive VC with automatic alignment search), MI355X-native drop-in for
`seq2seq_vc.models.AASVC` (reference models/aas_vc.py:39-603): same constructor keywords, `forward`
returning the same dict, `inference`, state_dict keys, and the `viterbi_func` slot (:132).

Hot-path differences that do not change results: the pairwise-distance tensor (B,T_f,T_x,adim) is never
materialised; the alignment search, duration bincount and binarisation loss run on the GPU in one
launch (no per-utterance device->host->device hop); masks are length vectors.
"""
import os

import torch
from torch import nn

from .. import modules as Mo
from ..conformer import ConformerEncoder
from ..ops import functional as Fn
from ..ops import functional_aas as FA
from ..sdp import DurationPredictor, StochasticDurationPredictor

MAX_DP_OUTPUT = 10
_FBRANCH = os.environ.get("S2SVC_AAS_FBRANCH", "1") != "0"      # A/B aid: the alignment module's feature side on the auxiliary stream


class AlignmentModule(nn.Module):
    """Alignment learning framework (reference modules/alignments.py:12-60)."""

    def __init__(self, adim, odim):
        super().__init__()
        self.t_conv1 = nn.Conv1d(adim, adim, kernel_size=3, padding=1)
        self.t_conv2 = nn.Conv1d(adim, adim, kernel_size=1, padding=0)
        self.f_conv1 = nn.Conv1d(odim, adim, kernel_size=3, padding=1)
        self.f_conv2 = nn.Conv1d(adim, adim, kernel_size=3, padding=1)
        self.f_conv3 = nn.Conv1d(adim, adim, kernel_size=1, padding=0)

    def feats_branch(self, feats, feat_lens=None):
        """The acoustic-feature side (f_conv1..3): depends on the target features only, so the training forward pass runs it on
        the auxiliary stream beside the encoder (AASVC._forward) -- and autograd then runs its backward pass there too.
        feat_lens: Lens of feats; in a captured step its crop() keeps the frames beyond the batch's longest utterance out of the taps
        (the features themselves are zero there: batch padding)."""
        vl = Mo.crop_dev(feat_lens)
        f = Fn.conv1d(feats, self.f_conv1.weight, self.f_conv1.bias, act="relu")
        f = Fn.conv1d(f, self.f_conv2.weight, self.f_conv2.bias, act="relu", vlens=vl)
        return Fn.linear(f, self.f_conv3.weight, self.f_conv3.bias)

    def forward(self, text, feats, text_lens=None, f=None, feat_lens=None):
        """text (B,T_text,adim), feats (B,T_feats,odim), text_lens: Lens -> log_p_attn (B,T_feats,T_text) fp32.
        f: the result of feats_branch(feats) if the caller has it already."""
        t = Fn.conv1d(text, self.t_conv1.weight, self.t_conv1.bias, act="relu", vlens=Mo.crop_dev(text_lens))
        t = Fn.linear(t, self.t_conv2.weight, self.t_conv2.bias)
        if f is None:
            f = self.feats_branch(feats, feat_lens)
        return FA.pairwise_logsoftmax(f, t, None if text_lens is None else text_lens.dev)


def viterbi_decode(log_p_attn, text_lengths, feats_lengths):
    """GPU replacement of modules/alignments.py:281-310: (ds (B,T_text) fp32, bin_loss scalar)."""
    dev = log_p_attn.device
    tl, fl = Mo.Lens.of(text_lengths, dev), Mo.Lens.of(feats_lengths, dev)
    ds, bin_loss, _ = Fn.viterbi_decode(log_p_attn, tl.dev, fl.dev)
    return ds, bin_loss


class GaussianUpsampling(nn.Module):
    def __init__(self, delta=0.1):
        super().__init__()
        self.delta = delta

    def forward(self, hs, ds, feat_lens, text_lens, T_feats):
        return FA.gaussian_upsample(hs, ds, None if text_lens is None else text_lens.dev,
                                    None if feat_lens is None else feat_lens.dev, T_feats, self.delta)


class AASVC(nn.Module):
    def __init__(self, idim, odim, adim: int = 384, aheads: int = 4, elayers: int = 6, eunits: int = 1536, dlayers: int = 6,
                 dunits: int = 1536, postnet_layers: int = 5, postnet_chans: int = 512, postnet_filts: int = 5,
                 positionwise_layer_type: str = "conv1d", positionwise_conv_kernel_size: int = 1,
                 use_scaled_pos_enc: bool = True, use_batch_norm: bool = True, encoder_input_layer: str = "linear",
                 encoder_input_conv_kernel_size: int = 3, encoder_normalize_before: bool = False,
                 decoder_normalize_before: bool = False, encoder_concat_after: bool = False,
                 decoder_concat_after: bool = False, duration_predictor_use_encoder_outputs: bool = True,
                 duration_predictor_input_dim: int = None, duration_predictor_layers: int = 2,
                 duration_predictor_chans: int = 384, duration_predictor_kernel_size: int = 3,
                 encoder_reduction_factor: int = 1, post_encoder_reduction_factor: int = 1,
                 decoder_reduction_factor: int = 1, encoder_type: str = "conformer", decoder_type: str = "conformer",
                 duration_predictor_type: str = "deterministic", conformer_pos_enc_layer_type: str = "rel_pos",
                 conformer_self_attn_layer_type: str = "rel_selfattn", use_macaron_style_in_conformer: bool = True,
                 use_cnn_in_conformer: bool = True, conformer_enc_kernel_size: int = 7, conformer_dec_kernel_size: int = 31,
                 spk_embed_dim: int = None, spk_embed_integration_type: str = "add",
                 transformer_enc_dropout_rate: float = 0.1, transformer_enc_positional_dropout_rate: float = 0.1,
                 transformer_enc_attn_dropout_rate: float = 0.1, transformer_dec_dropout_rate: float = 0.1,
                 transformer_dec_positional_dropout_rate: float = 0.1, transformer_dec_attn_dropout_rate: float = 0.1,
                 duration_predictor_dropout_rate: float = 0.1, postnet_dropout_rate: float = 0.5,
                 init_type: str = "xavier_uniform", use_masking: bool = False, use_weighted_masking: bool = False,
                 diffsinger_denoiser_residual_channels: int = 256, prodiff_denoiser_layers: int = 20,
                 prodiff_denoiser_channels: int = 256, prodiff_diffusion_steps: int = 1000,
                 prodiff_diffusion_timescale: int = 1, prodiff_diffusion_beta: float = 40.0,
                 prodiff_diffusion_scheduler: str = "vpsde", prodiff_diffusion_cycle_ln: int = 1,
                 stochastic_duration_predictor_kernel_size: int = 3,
                 stochastic_duration_predictor_dropout_rate: float = 0.5, stochastic_duration_predictor_flows: int = 4,
                 stochastic_duration_predictor_dds_conv_layers: int = 3,
                 stochastic_duration_predictor_noise_scale: float = 0.8):
        nn.Module.__init__(self)
        self.idim, self.odim = idim, odim
        if spk_embed_dim is not None:
            raise NotImplementedError("speaker-embedding integration is out of scope (no recipe config uses it)")
        self.spk_embed_dim = None
        self.encoder_reduction_factor = encoder_reduction_factor
        self.post_encoder_reduction_factor = post_encoder_reduction_factor
        self.decoder_reduction_factor = decoder_reduction_factor
        self.encoder_type, self.decoder_type = encoder_type, decoder_type
        self.duration_predictor_type = duration_predictor_type
        self.use_scaled_pos_enc = use_scaled_pos_enc
        self.encoder_input_layer = encoder_input_layer
        self.duration_predictor_use_encoder_outputs = duration_predictor_use_encoder_outputs
        self.viterbi_func = viterbi_decode
        # losses.ForwardSumLoss.prefetch of the criterion that will see `log_p_attn` (set by the trainers): the loss's dependent
        # recursion then runs on the auxiliary stream beside the decoder
        self.forward_sum_prefetch = None
        self.stochastic_duration_predictor_noise_scale = stochastic_duration_predictor_noise_scale
        if encoder_type != "conformer" or decoder_type != "conformer":
            raise NotImplementedError("only the conformer encoder/decoder of the vc2 recipes is supported")

        self.encoder = ConformerEncoder(
            idim=idim * encoder_reduction_factor, attention_dim=adim, attention_heads=aheads, linear_units=eunits,
            num_blocks=elayers, input_layer=encoder_input_layer, dropout_rate=transformer_enc_dropout_rate,
            positional_dropout_rate=transformer_enc_positional_dropout_rate,
            attention_dropout_rate=transformer_enc_attn_dropout_rate, normalize_before=encoder_normalize_before,
            concat_after=encoder_concat_after, positionwise_layer_type=positionwise_layer_type,
            positionwise_conv_kernel_size=positionwise_conv_kernel_size, macaron_style=use_macaron_style_in_conformer,
            pos_enc_layer_type=conformer_pos_enc_layer_type, selfattention_layer_type=conformer_self_attn_layer_type,
            use_cnn_module=use_cnn_in_conformer, cnn_module_kernel=conformer_enc_kernel_size)
        if duration_predictor_type == "deterministic":
            self.duration_predictor = DurationPredictor(idim=adim, n_layers=duration_predictor_layers,
                                                        n_chans=duration_predictor_chans,
                                                        kernel_size=duration_predictor_kernel_size,
                                                        dropout_rate=duration_predictor_dropout_rate)
        elif duration_predictor_type == "stochastic":
            self.duration_predictor = StochasticDurationPredictor(
                channels=adim, kernel_size=stochastic_duration_predictor_kernel_size,
                dropout_rate=stochastic_duration_predictor_dropout_rate, flows=stochastic_duration_predictor_flows,
                dds_conv_layers=stochastic_duration_predictor_dds_conv_layers, global_channels=-1)
        else:
            raise ValueError(f"Duration predictor type: {duration_predictor_type} is not supported.")
        if not self.duration_predictor_use_encoder_outputs:
            self.duration_predictor_projection = Mo.Conv2dSubsampling(duration_predictor_input_dim, adim, 0.0, use_pos_enc=False)
        self.alignment_module = AlignmentModule(adim * post_encoder_reduction_factor, odim * decoder_reduction_factor)
        self.length_regulator = GaussianUpsampling()
        self.decoder = ConformerEncoder(
            idim=0, attention_dim=adim * post_encoder_reduction_factor, attention_heads=aheads, linear_units=dunits,
            num_blocks=dlayers, input_layer=None, dropout_rate=transformer_dec_dropout_rate,
            positional_dropout_rate=transformer_dec_positional_dropout_rate,
            attention_dropout_rate=transformer_dec_attn_dropout_rate, normalize_before=decoder_normalize_before,
            concat_after=decoder_concat_after, positionwise_layer_type=positionwise_layer_type,
            positionwise_conv_kernel_size=positionwise_conv_kernel_size, macaron_style=use_macaron_style_in_conformer,
            pos_enc_layer_type=conformer_pos_enc_layer_type, selfattention_layer_type=conformer_self_attn_layer_type,
            use_cnn_module=use_cnn_in_conformer, cnn_module_kernel=conformer_dec_kernel_size)
        self.feat_out = nn.Linear(adim * post_encoder_reduction_factor, odim * decoder_reduction_factor)
        self.postnet = None if postnet_layers == 0 else Mo.Postnet(
            idim=idim, odim=odim, n_layers=postnet_layers, n_chans=postnet_chans, n_filts=postnet_filts,
            use_batch_norm=use_batch_norm, dropout_rate=postnet_dropout_rate)
        self.decoder.cut_name = "decoder"

    def dp_plan(self):
        """Stages of the data-parallel backward pass (distributed.OverlappedBackward).  The decoder holds 113 M of the 157 M
        parameters of the vc2 configuration (4 layers x 28 M at d = 1536).  Two loss keys: "decoder" (the L1 loss: reaches the
        postnet / decoder / length regulator) and "align" (forward-sum + binarisation + duration losses: reach the alignment
        module and the duration predictor); below the cut at the encoder output both meet and the encoder runs last.
        Stage 1 runs BOTH roots: first "align", rooted on the auxiliary stream (`branch_root`, ops.functional.branch_backward:
        the duration branch -- input projection + predictor -- ran there in the forward pass, so its whole backward pass runs
        there; the main stream only gets the short alignment-module part), then "decoder" down to the encoder output on the
        calling stream, beside it; the stage ends with the join.  Stage 2 is the encoder.

        Round 4: the stochastic predictor's two conditioning networks run in the LAST stage (cut "sdp_cond", see below).

        Why two stages and not one per decoder layer (round 2's plan: 3 of 4 layers in stage 1, `dp_decoder_stages = 1`): the duration
        branch's backward pass is ~350 small dependent launches that take 4.4 ms beside the decoder's GEMMs, a decoder layer's
        backward pass 1.2 ms -- with fewer than four layers beside it the branch is the critical path of the stage (measured on
        one MI355X, `bench.py --split-backward --stage-times`: stage graphs 10.97 + 2.04 ms with h = 0, 10.80 + 1.20 + 2.04 with
        h = 1, 10.28 + 1.17 + 1.26 + 2.02 with h = 2: every layer moved out of stage 1 adds ~1 ms to the step).  Buckets with two
        stages: 568 | 62 MB fp32 (half with the opt-in bf16 payload); the first travels behind the encoder's 1.9 ms -- a ring
        all-reduce of S bytes over N ranks takes 2 (N - 1) / N x S / bus bandwidth, i.e. 3.3 ms at 300 GB/s and N = 8, of which
        ~ 1.4 ms stay exposed (round 5, DESIGN.md section 7: the model and the plans with decoder cuts, which cost 0.45-0.8 ms per cut
        on one GPU and come out even at 300 GB/s) --, the second is exposed (as the last bucket of any plan is)."""
        dec = list(self.decoder.encoders)
        tail = [m for m in (getattr(self.decoder, "after_norm", None), self.feat_out, self.postnet) if m is not None]
        side = [self.alignment_module, self.duration_predictor]
        if hasattr(self, "duration_predictor_projection"):
            side.append(self.duration_predictor_projection)
        h = int(getattr(self, "dp_decoder_stages", 0))        # decoder layers that get a stage of their own (trainer config `dp_decoder_stages`)
        h = max(0, min(h, len(dec) - 1))
        last = {"root": "cut:encoder_out", "modules": [self.encoder]}
        sdp = self.duration_predictor
        # 0: no cut, 1: the network behind x, 2: both conditioning networks.  bench.py --workload aasvc --split-backward, one box, ms per
        # staged step (stage graphs): 12.28 (9.49 + 1.45) / 12.00 (9.18 + 1.73) / 11.76-11.85 (8.87 + 1.91) against 11.46 for the one-graph step
        sdp_cut = "2"
        if self.duration_predictor_type == "stochastic" and sdp_cut != "0":
            # Round 4: the duration branch's backward pass (~ 4 ms of small launches on the auxiliary stream) had become the critical path
            # of stage 1 once the decoder's backward pass beside it got shorter; its last part -- the two conditioning networks
            # below the "sdp_cond" cut (sdp.py) -- now runs in the LAST stage, on the auxiliary stream beside the encoder's backward
            # pass, and their parameters travel with the encoder's bucket.
            cond = [sdp.pre, sdp.dds, sdp.proj] + ([sdp.post_pre, sdp.post_dds, sdp.post_proj] if sdp_cut == "2" else [])
            inner = {id(p) for m in cond for p in m.parameters()}
            side = [m for m in side if m is not sdp] + [p for p in sdp.parameters() if id(p) not in inner]
            last = {"root": "cut:encoder_out", "branch_root": "cut:sdp_cond" if sdp_cut == "2" else "cut:sdp_cond_x", "modules": [self.encoder] + cond}
        plan = [{"root": "loss:decoder", "branch_root": "loss:align", "modules": dec[h:] + tail + side}]
        for li in range(h, 0, -1):
            plan.append({"root": f"cut:decoder.{li}", "modules": [dec[li - 1]]})
        plan.append(last)
        return plan

    # ---------------------------------------------------------------------------------------------
    def _forward(self, xs, ilens, ys=None, olens=None, dp_inputs=None, dplens=None, spembs=None, is_inference=False):
        ret = {}
        dev = xs.device
        il = Mo.Lens.of(ilens, dev)
        ol = Mo.Lens.of(olens, dev) if olens is not None else None
        er, pr, dr = self.encoder_reduction_factor, self.post_encoder_reduction_factor, self.decoder_reduction_factor
        if er > 1:
            b, tmax, dim = xs.shape
            if tmax % er != 0:
                xs = xs[:, : -(tmax % er)]
            xs = xs.contiguous().view(b, tmax // er, dim * er)
            il = il.map(lambda v: v // er)
        olr = ol
        if dr > 1 and ys is not None:
            b, tmax, dim = ys.shape
            if tmax % dr != 0:
                ys = ys[:, : -(tmax % dr)]
            ys = ys.contiguous().view(b, tmax // dr, dim * dr)
            olr = ol.map(lambda v: v // dr)
        f_pre = None
        if not is_inference and ys is not None and xs.is_cuda and _FBRANCH:
            # the alignment module's feature side needs nothing from the encoder: auxiliary stream, beside it (forward and backward)
            f_pre = Fn.branch_run(lambda: self.alignment_module.feats_branch(Fn.to_compute(ys), olr), uses=(ys,))
        hs, _ = self.encoder(Fn.to_compute(xs), il)
        hs = Fn.cut_point(hs, "encoder_out")
        if self.encoder_input_layer == "conv2d":
            il = il.map(lambda v: ((v - 2 + 1) // 2 - 2 + 1) // 2)
        if pr > 1:
            b, tmax, dim = hs.shape
            if tmax % pr != 0:
                hs = hs[:, : -(tmax % pr)]
            hs = hs.contiguous().view(b, tmax // pr, dim * pr)
            il = il.map(lambda v: v // pr)
        Th = hs.shape[1]
        uses_hs = self.duration_predictor_use_encoder_outputs and self.duration_predictor_type != "stochastic"     # (the stochastic one detaches)
        hs_dp = hs
        if not is_inference and ys is not None:
            # consumers of the encoder output: alignment module, length regulator (, deterministic duration predictor): one launch sums
            # their gradients (Fn.fan_out) instead of autograd's element-wise adds
            hs, hs_al, *rest = Fn.fan_out(hs, 3 if uses_hs else 2)
            hs_dp = rest[0] if rest else hs.detach()
        else:
            hs_al = hs

        def dp_input():
            """Input of the duration predictor: the encoder output or the projection of `dp_inputs`.  The projection belongs to
            the duration branch: in training it runs inside branch_run, so that no node of the branch's backward pass sits on
            the main stream (where it would stall everything queued behind it until the branch has finished)."""
            if self.duration_predictor_use_encoder_outputs:
                return hs_dp
            if dplens is not None and Mo._BANK is not None:
                # captured step: the projection's output and the encoder's have padded lengths; F.interpolate's ratio is the one
                # of the reference's cropped tensors (graph data, modules.LensBank)
                d, dl = self.duration_predictor_projection(Fn.to_compute(dp_inputs), Mo.Lens.of(dplens, dev))
                return FA.interp_nearest(d, Th, Mo.crop_dev(dl), Mo.crop_dev(il))
            d, _ = self.duration_predictor_projection(Fn.to_compute(dp_inputs), None)
            return FA.interp_nearest(d, Th)

        stochastic_training = (not is_inference) and self.duration_predictor_type == "stochastic"
        dpi = None if stochastic_training else dp_input()
        Tx = hs.shape[1]
        il_c = il.clamp(Tx)
        stochastic = self.duration_predictor_type == "stochastic"
        if is_inference:
            log_p_attn, ds, bin_loss = None, None, 0.0
            if ys is not None:
                log_p_attn = self.alignment_module(hs, Fn.to_compute(ys), il_c)
                ds, bin_loss = self.viterbi_func(log_p_attn, il_c, olr)
            if stochastic:
                d_outs = self.duration_predictor.forward_cl(dpi, il_c, inverse=True,
                                                            noise_scale=self.stochastic_duration_predictor_noise_scale)
            else:
                d_outs = self.duration_predictor.inference(dpi, None)
            d_outs = torch.clamp(d_outs, max=MAX_DP_OUTPUT)
            ret["d_outs"] = d_outs
            dsf = d_outs.float()
            if float(dsf.sum()) == 0:
                dsf = dsf.clone()
                dsf[dsf.sum(dim=1).eq(0)] = 1
            T_feats = int(dsf.sum())
            hs = self.length_regulator(hs, dsf, None, il_c, T_feats)
            dec_lens = None
        else:
            if f_pre is not None:
                Fn.branch_join(f_pre)
                log_p_attn = self.alignment_module(hs_al, None, il_c, f=f_pre)
            else:
                log_p_attn = self.alignment_module(hs_al, Fn.to_compute(ys), il_c, feat_lens=olr)
            # two consumers: the alignment search's binarisation loss here, the forward-sum loss of the trainer (returned tensor)
            log_p_attn, lp_search = Fn.fan_out(log_p_attn, 2)
            if self.forward_sum_prefetch is not None:
                self.forward_sum_prefetch(log_p_attn, il, olr)
            ds, bin_loss = self.viterbi_func(lp_search, il_c, olr)
            if stochastic:
                # ~330 small launches that depend on nothing the length regulator / decoder / postnet below produce: they
                # run on the auxiliary stream beside them (forward here, backward through autograd's stream rule)
                # (normalize: / the number of non-pad text positions, aas_vc.py:403, inside the NLL kernel)
                ret["dur_nll"] = Fn.branch_run(lambda: self.duration_predictor.forward_cl(dp_input(), il_c, w=ds, normalize=True),
                                               uses=(ds, hs, dp_inputs, il_c.dev))
            else:
                d_outs = self.duration_predictor(dpi, il_c)
                ret["d_outs"] = torch.clamp(d_outs, max=MAX_DP_OUTPUT)
            hs = self.length_regulator(hs, ds, olr, il_c, olr.max())
            dec_lens = olr
        zs, _ = self.decoder(hs, dec_lens)
        before = Fn.linear(zs, self.feat_out.weight, self.feat_out.bias).view(zs.size(0), -1, self.odim)
        b_res = b_post = before
        if self.postnet is not None:        # three consumers (loss, residual, Postnet): their gradients meet in one launch
            before, b_res, b_post = Fn.fan_out(before, 3)
        post_lens = None
        if self.postnet is not None and dec_lens is not None and dec_lens.cap is not None:     # captured step: frames of `before`
            post_lens = dec_lens if dr == 1 else dec_lens.map(lambda v, _r=dr: v * _r)
        after = before if self.postnet is None else Fn.add_dropout(b_res, self.postnet(b_post, post_lens), 0.0)
        ret["before_outs"], ret["after_outs"] = before, after
        Fn.branch_join(ret.get("dur_nll"), *(getattr(log_p_attn, "_s2s_fs", None) or ())[:2])
        ret["ds"] = ds
        ret["ilens"] = Mo.tag_lens(self._lens_like(ilens, il.host), il)
        ret["bin_loss"] = bin_loss
        ret["log_p_attn"] = log_p_attn
        ret["olens_reduced"] = Mo.tag_lens(self._lens_like(olens, olr.host), olr) if olr is not None else None
        return ret

    @staticmethod
    def _lens_like(proto, values):
        if isinstance(proto, torch.Tensor):
            return proto.new_tensor(list(values))
        return torch.tensor(list(values))

    def forward(self, src_speech, src_speech_lengths, tgt_speech, tgt_speech_lengths, dp_inputs=None, dp_lengths=None,
                spembs=None):
        """-> dict(before_outs, after_outs, ds, ilens, bin_loss, log_p_attn, olens_reduced, dur_nll | d_outs, olens, ys)."""
        dev = src_speech.device
        il, ol = Mo.Lens.of(src_speech_lengths, dev), Mo.Lens.of(tgt_speech_lengths, dev)
        xs, ys = src_speech[:, : il.max()], tgt_speech[:, : ol.max()]
        ret = self._forward(xs, src_speech_lengths, ys, tgt_speech_lengths, dp_inputs=dp_inputs, dplens=dp_lengths,
                            spembs=spembs, is_inference=False)
        olens = tgt_speech_lengths
        if self.decoder_reduction_factor > 1:
            ol_r = ol.map(lambda v, _r=self.decoder_reduction_factor: v - v % _r)
            olens = Mo.tag_lens(self._lens_like(tgt_speech_lengths, ol_r.host), ol_r)
            ys = ys[:, : ol_r.max()]
        ret["olens"], ret["ys"] = olens, ys
        return ret

    @torch.no_grad()
    def inference(self, src_speech, tgt_speech=None, spembs=None, dp_input=None, use_teacher_forcing=False):
        x, y = src_speech, tgt_speech
        ilens = torch.tensor([x.shape[0]], dtype=torch.long)
        ys = y.unsqueeze(0) if y is not None else None
        olens = torch.tensor([y.shape[0]], dtype=torch.long) if y is not None else None
        if use_teacher_forcing:
            raise NotImplementedError("teacher-forced inference is broken in the reference (models/aas_vc.py:572) and unused")
        ret = self._forward(x.unsqueeze(0), ilens, ys=ys, olens=olens, dp_inputs=None if dp_input is None else dp_input.unsqueeze(0),
                            is_inference=True)
        outs, d_outs = ret["after_outs"], ret["d_outs"]
        if ret["ds"] is None and ret["log_p_attn"] is None:
            return outs[0].float(), d_outs[0]
        return outs[0].float(), d_outs[0], ret["ds"][0], ret["log_p_attn"][0], ret["ilens"][0]
