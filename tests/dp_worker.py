"""One rank of a data-parallel trainer run (spawned by gpu_model_check.dp_trainers_two_ranks).

    python tests/dp_worker.py <vtn|aasvc> <rank> <world> <port> <out.pt> [payload] [none|trace|graph] [allreduce|rs_ag] [stages|flush]

The last argument runs the trainer with config["hip_graph"] ("trace": the eager reference of the captured step, "graph":
stage graphs replayed with the all-reduces between them; trainers/graphed.py), 5 steps through Trainer._step.

Both ranks share the one GPU of the box and talk over gloo (RCCL refuses two ranks on one device); the product code path
is the same one RCCL runs on a multi-GPU node: Trainer(config["distributed"]) -> broadcast of rank 0's parameters ->
distributed.OverlappedBackward (gradient cuts, stage-by-stage backward, one asynchronous all-reduce per stage).
Rank r trains on its share of the golden batch; rank 1 deliberately starts from DIFFERENT weights, so the run only matches
the single-process replay if the initial broadcast happened."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shares(kind, z, rank, world):
    t = lambda k: torch.from_numpy(z[k])
    B = z["in.xs"].shape[0]
    cut = [(B * r + world - 1) // world for r in range(world + 1)]          # ceil split: rank 0 gets the larger share
    cut[-1] = B
    sl = slice(cut[rank], cut[rank + 1])
    if kind == "vtn":
        return {"xs": t("in.xs")[sl], "ilens": t("in.ilens")[sl], "ys": t("in.ys")[sl], "labels": t("in.labels")[sl],
                "olens": t("in.olens")[sl]}, sl
    return {"xs": t("in.xs")[sl], "ilens": t("in.ilens")[sl], "ys": t("in.ys")[sl], "olens": t("in.olens")[sl],
            "dp_inputs": t("in.xs")[sl], "dplens": t("in.ilens")[sl]}, sl


def set_noise(kind, model, z, cfg, batch, sl):
    """The stochastic duration predictor consumes an injected draw once per forward pass: (B_share, 2, T_text of the share)."""
    if kind == "aasvc":
        red = cfg.get("encoder_reduction_factor", 1) * cfg.get("post_encoder_reduction_factor", 1)
        model.duration_predictor.noise = torch.from_numpy(z["in.sdp_noise"])[sl][:, :, : int(batch["ilens"].max()) // red].contiguous()


def build(kind, z, cfg, perturb=False):
    from seq2seq_vc_amd import losses as L
    from seq2seq_vc_amd import models as M
    import gpu_model_check as mc
    model = (M.VTN if kind == "vtn" else M.AASVC)(**mc.model_cfg(cfg))
    model.load_state_dict(mc.sd_of(z))
    if perturb:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.5)
    model.to("cuda").train()
    mc._kill_dropout(model)
    if kind == "vtn":
        crit = {"Seq2SeqLoss": L.Seq2SeqLoss(10.0)}
        conf = {"train_max_steps": 10 ** 9, "log_interval_steps": 1, "save_interval_steps": 10 ** 9, "grad_norm": 1.0, "outdir": ".",
                "side_streams": 0, "inline_batches": False}
    else:
        crit = {"L1Loss": L.L1Loss(), "ForwardSumLoss": L.ForwardSumLoss()}
        conf = {"train_max_steps": 10 ** 9, "log_interval_steps": 1, "save_interval_steps": 10 ** 9, "grad_norm": 1.0, "outdir": ".",
                "criterions": ["L1Loss", "ForwardSumLoss", "StochasticDurationPredictorLoss"], "lambda_align": 2.0,
                "dp_train_start_steps": -1, "side_streams": 0, "inline_batches": False}
    return model, crit, conf


def main():
    kind, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    payload = sys.argv[6] if len(sys.argv) > 6 else "fp32"
    graph_mode = sys.argv[7] if len(sys.argv) > 7 else "none"
    collective = sys.argv[8] if len(sys.argv) > 8 else "allreduce"
    exchange = sys.argv[9] if len(sys.argv) > 9 else "stages"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import gpu_model_check as mc
    from seq2seq_vc_amd import trainers as T
    from seq2seq_vc_amd.ops import functional as Fn
    from seq2seq_vc_amd.ops import kernels as K
    from seq2seq_vc_amd.optim import FlatAdam
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Fn.set_compute_dtype(torch.float32)
    K.manual_seed(7)
    cfg, z = mc.load("vtn_tiny_train" if kind == "vtn" else "aasvc_tiny_train")
    model, crit, conf = build(kind, z, cfg, perturb=(rank != 0))
    conf = dict(conf, distributed=True, rank=rank, dp_grad_payload=payload, dp_collective=collective, dp_exchange=exchange)
    if exchange == "flush":        # gradient batches of 4 closures (VTN: forked to two side streams): several flushes = several buckets
        conf.update(side_streams=2 if kind == "vtn" else 0, inline_batches=kind != "vtn", gradient_batch=4, dp_min_bucket_mb=0.05)
    if graph_mode != "none":
        conf.update(hip_graph=(True if graph_mode == "graph" else "trace"), graph_length_quantum=8)
    else:
        conf.update(hip_graph=False)
    opt = FlatAdam(model, lr=1e-3, grad_norm=1.0, warmup_steps=10)
    batch, sl = shares(kind, z, rank, world)
    cls = T.ARVCTrainer if kind == "vtn" else T.AASVCTrainer
    tr = cls(0, 0, {"train": [batch] * 6}, None, model, None, crit, opt, None, conf, device="cuda")
    logs = []
    tr.log_fn = lambda step, d: logs.append(dict(d))
    if graph_mode != "none":
        if kind == "aasvc":          # one fixed draw per shape, device-resident (the injected draw is consumed by a host copy)
            noise, gen = {}, torch.Generator().manual_seed(5)

            def fixed_noise(shape, device):
                if tuple(shape) not in noise:
                    noise[tuple(shape)] = torch.randn(shape, generator=gen).to(device)
                return noise[tuple(shape)]

            model.duration_predictor._randn = fixed_noise
        for _ in range(5):
            tr._step(batch)
            tr._check_log_interval()
    else:
        for _ in range(3):
            set_noise(kind, model, z, cfg, batch, sl)
            tr._train_step(batch)
            tr._check_log_interval()
    torch.cuda.synchronize()
    torch.save({"flat_p": opt.flat_p.detach().cpu(), "buffers": {k: v.detach().cpu() for k, v in model.named_buffers()},
                "logs": logs, "stages": len(tr.dp.plan) if tr.dp is not None else 0, "steps": tr.steps,
                "fx_buckets": [] if tr.fx is None else tr.fx.bucket_bytes(), "fx_flushes": 0 if tr.fx is None else tr.fx.n_flushes,
                "graphs": 0 if tr._graphed is None else sum(len(e.graphs) for e in tr._graphed.entries.values()),
                "bucket_bytes": tr.dp.bucket_bytes() if tr.dp is not None else []}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
