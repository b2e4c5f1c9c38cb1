#!/usr/bin/env python3
"""AAS-VC training-step benchmark (configuration C3; one rank's share).

    python tools/bench_aasvc.py [--batch 16] [--steps 20] [--warmup 3] [--dtype bf16] [--no-graph]

Model: egs/arctic/vc2/conf/aas_vc.melmelmel.v1.yaml (Conformer 4+4, d=384, stochastic duration predictor),
batch 16 utterance pairs, T_src = T_tgt padded to 256, bf16 compute with fp32 master weights.
One step = forward (encoder, alignment module + MAS, duration predictor, Gaussian upsampling, decoder,
postnet) + L1 + lambda_align*(forward-sum + bin) + duration NLL + backward + clip + Adam + WarmupLR
(trainers/aas_vc.py:56-164).  MAS runs on the device, so the whole step is captured as a hipGraph.
Prints ONE JSON line (mel-frames/sec = valid target frames per step / step time).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AASVC_VC2 = dict(
    idim=80, odim=80, adim=384, aheads=2, elayers=4, eunits=1536, dlayers=4, dunits=1536, positionwise_layer_type="linear",
    positionwise_conv_kernel_size=1, duration_predictor_use_encoder_outputs=False, duration_predictor_input_dim=80,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3, postnet_layers=5,
    postnet_filts=5, postnet_chans=256, use_masking=True, encoder_normalize_before=True, decoder_normalize_before=True,
    encoder_reduction_factor=1, post_encoder_reduction_factor=4, decoder_reduction_factor=1, encoder_type="conformer",
    decoder_type="conformer", duration_predictor_type="stochastic", encoder_input_layer="linear",
    conformer_pos_enc_layer_type="rel_pos", conformer_self_attn_layer_type="rel_selfattn",
    use_macaron_style_in_conformer=True, use_cnn_in_conformer=True, conformer_enc_kernel_size=15, conformer_dec_kernel_size=15,
    init_type="xavier_uniform", transformer_enc_dropout_rate=0.2, transformer_enc_positional_dropout_rate=0.2,
    transformer_enc_attn_dropout_rate=0.2, transformer_dec_dropout_rate=0.2, transformer_dec_positional_dropout_rate=0.2,
    transformer_dec_attn_dropout_rate=0.2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--side-streams", type=int, default=0, help="0 (default): gradient work stays on its stream, batched; n > 0: forked to n side streams")
    ap.add_argument("--no-inline-batches", action="store_true")
    a = ap.parse_args()
    from bench import canonical_batch
    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd.models import AASVC
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.optim import FlatAdam
    dev = torch.device("cuda", 0)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    Fn.set_compute_dtype(dtype)
    Fn.enable_side_streams(a.side_streams, inline_batches=not a.no_inline_batches)
    K.manual_seed(1234)
    xs, ilens, ys, _, olens = canonical_batch(a.batch)
    xs_d, ys_d = xs.to(dev), ys.to(dev)
    torch.manual_seed(0)
    model = AASVC(**AASVC_VC2).to(dev)
    model.train()
    l1_crit, fs_crit = L.L1Loss(), L.ForwardSumLoss()
    opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=(dtype == torch.bfloat16))
    loss_buf = torch.zeros(4, device=dev)
    lambda_align = 2.0

    def fwd_bwd():
        K.reset_op_counter()
        K.advance_seed(dev)
        ret = model(xs_d, ilens, ys_d, olens, xs_d, dp_lengths=ilens)
        l1 = l1_crit(ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
        fs = fs_crit(ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
        dur = torch.sum(ret["dur_nll"].float())
        loss = l1 + lambda_align * (fs + ret["bin_loss"]) + dur
        loss.backward()
        Fn.side_join()
        loss_buf.copy_(torch.stack([l1.detach().float(), fs.detach().float(), ret["bin_loss"].detach().float(), dur.detach()]))

    def step_eager():
        fwd_bwd()
        opt.step()
        opt.zero_grad()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step_eager()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if not a.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step_eager()
        except Exception as e:  # noqa: BLE001 -- say so and run eagerly
            print(f"[bench_aasvc] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    step = graph.replay if graph is not None else step_eager
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / a.steps
    lb = loss_buf.tolist()
    print(json.dumps({"metric": "mel-frames/sec (train)", "value": float(olens.sum()) / t, "unit": "mel-frames/sec", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": t * 1e3, "higher_is_better": True, "dtype": a.dtype,
                      "data": "synthetic", "config": {"workload": "AAS-VC egs/arctic/vc2 (aas_vc.melmelmel.v1.yaml) training step",
                                                      "batch_per_gpu": a.batch, "T_src": 256, "T_tgt": 256,
                                                      "hip_graph": graph is not None,
                                                      "params_M": sum(p.numel() for p in model.parameters()) / 1e6},
                      "final_losses": {"l1": lb[0], "forward_sum": lb[1], "bin": lb[2], "dur_nll": lb[3],
                                       **opt.last_stats()}}))


if __name__ == "__main__":
    main()
