"""Full-size soak of the captured trainer steps (config["hip_graph"], trainers/graphed.py): N steps over batches of three padded
shapes and ever-changing lengths, bf16, dropout on -- the trainer that replays hipGraphs must end with the parameters of the one
that runs the same padded batches eagerly ("trace"), bit for bit.

    python tools/soak_trainer.py vtn 90 ; python tools/soak_trainer.py aasvc 45
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from seq2seq_vc_amd import losses as L, models as M, trainers as T
from seq2seq_vc_amd.ops import functional as Fn, kernels as K
from seq2seq_vc_amd.optim import FlatAdam
dev = torch.device("cuda", 0)
dtype = torch.bfloat16


def batches(wlname, n, B, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for k in range(n):
        hi_in, hi_out = [(256, 256), (192, 256), (256, 192)][k % 3]
        ilens = torch.randint(hi_in - 60, hi_in + 1, (B,), generator=g)
        olens = torch.randint(hi_out - 60, hi_out + 1, (B,), generator=g)
        Ti, To = int(ilens.max()), int(olens.max())
        xs, ys = torch.randn(B, Ti, 80, generator=g), torch.randn(B, To, 80, generator=g)
        for b in range(B):
            xs[b, ilens[b]:] = 0
            ys[b, olens[b]:] = 0
        bt = {"xs": xs, "ilens": ilens, "ys": ys, "olens": olens}
        if wlname == "vtn":
            lab = torch.zeros(B, To)
            for b in range(B):
                lab[b, olens[b] - 1:] = 1.0
            bt["labels"] = lab
        else:
            bt["dp_inputs"], bt["dplens"] = xs, ilens
        out.append(bt)
    return out


def run(wlname, mode, data):
    Fn.set_compute_dtype(dtype)
    K.manual_seed(1234)
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    conf = {"train_max_steps": len(data), "log_interval_steps": 10, "save_interval_steps": 10 ** 9, "grad_norm": 1.0, "outdir": ".",
            "hip_graph": mode}
    if wlname == "vtn":
        model = M.VTN(**bench.VTN_VC1).to(dev).train()
        opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=True)
        tr = T.ARVCTrainer(0, 0, {"train": data}, None, model, None, {"Seq2SeqLoss": L.Seq2SeqLoss(10.0)}, opt, None, conf, device=dev)
    else:
        model = M.AASVC(**bench.AASVC_VC2).to(dev).train()
        noise, gen = {}, torch.Generator().manual_seed(5)

        def fixed(shape, device):          # one noise draw per shape for the duration predictor, made outside any capture
            if tuple(shape) not in noise:
                noise[tuple(shape)] = torch.randn(shape, generator=gen).to(device)
            return noise[tuple(shape)]

        model.duration_predictor._randn = fixed
        opt = FlatAdam(model, lr=8e-5, grad_norm=1.0, warmup_steps=4000, bf16_shadow=True)
        conf.update({"criterions": ["L1Loss", "ForwardSumLoss", "StochasticDurationPredictorLoss"], "lambda_align": 2.0,
                     "dp_train_start_steps": 0})
        tr = T.AASVCTrainer(0, 0, {"train": data}, None, model, None, {"L1Loss": L.L1Loss(), "ForwardSumLoss": L.ForwardSumLoss()},
                            opt, None, conf, device=dev)
    logs = []
    tr.log_fn = lambda s, d: logs.append(dict(d))
    tr.run()
    torch.cuda.synchronize()
    ng = sum(len(e.graphs) for e in tr._graphed.entries.values())
    return opt.flat_p.detach().clone(), logs, ng


def soak(wlname, steps):
    """-> dict(equal, max_diff, finite, graphs, logs_trace, logs_graph)"""
    data = batches(wlname, steps, 32 if wlname == "vtn" else 16)
    try:
        pt, lt, _ = run(wlname, "trace", data)
        pg, lg, ng = run(wlname, True, data)
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.enable_side_streams(0)
    return {"equal": bool(torch.equal(pt, pg)), "max_diff": float((pt - pg).abs().max()), "finite": bool(torch.isfinite(pg).all()),
            "graphs": ng, "logs_trace": lt, "logs_graph": lg}


if __name__ == "__main__":
    name, n = sys.argv[1], int(sys.argv[2])
    r = soak(name, n)
    print(name, "steps", n, "graphs", r["graphs"], "params equal", r["equal"], "max diff", r["max_diff"], "finite", r["finite"])
    print("last logs trace", {k: round(v, 5) for k, v in r["logs_trace"][-1].items()})
    print("last logs graph", {k: round(v, 5) for k, v in r["logs_graph"][-1].items()})
