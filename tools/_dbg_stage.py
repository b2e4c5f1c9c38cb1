import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_model_check as mc
from seq2seq_vc_amd import models as M, losses as L
from seq2seq_vc_amd.ops import functional as Fn, kernels as K
from seq2seq_vc_amd.optim import FlatAdam
from seq2seq_vc_amd.distributed import OverlappedBackward
NAME = os.environ.get("DBG_FIXTURE", "aasvc_tiny_train")
cfg, z = mc.load(NAME)
Fn.set_compute_dtype(torch.float32)
Fn.enable_side_streams(0)
model = M.AASVC(**mc.model_cfg(cfg)); model.load_state_dict(mc.sd_of(z)); model.to("cuda").train(); mc._kill_dropout(model)
opt = FlatAdam(model, lr=1e-3)
t = lambda k: torch.from_numpy(z[k])
xs, ys = t("in.xs").cuda(), t("in.ys").cuda()
il, ol = t("in.ilens"), t("in.olens")
noise = t("in.sdp_noise").cuda() if "in.sdp_noise" in z.files else None
from seq2seq_vc_amd.losses import DurationPredictorLoss
l1c, fsc = L.L1Loss(), L.ForwardSumLoss()
def fwd():
    if noise is not None:
        model.duration_predictor.noise = noise.clone()
    ret = model(xs, il, ys, ol, xs, dp_lengths=il)
    l1 = l1c(ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
    fs = fsc(ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
    if os.environ.get("DBG_ONLY") == "fs":
        return {"decoder": l1, "align": 2.0 * fs}
    if os.environ.get("DBG_ONLY") == "bin":
        return {"decoder": l1, "align": 2.0 * ret["bin_loss"]}
    if os.environ.get("DBG_ONLY") == "dur":
        return {"decoder": l1, "align": torch.sum(ret["dur_nll"].float())}
    dur = torch.sum(ret["dur_nll"].float()) if "dur_nll" in ret else DurationPredictorLoss()(ret["d_outs"], ret["ds"], ret["ilens"])
    return {"decoder": l1, "align": 2.0 * (fs + ret["bin_loss"]) + dur}
opt.zero_grad(); ls = fwd(); (ls["decoder"] + ls["align"]).backward(); Fn.side_join(); g_ref = opt.flat_g.clone()
ob = OverlappedBackward(model, opt, None, 1)
held = {}
def stage(i):
    if i == 0:
        opt.zero_grad()
        with ob.forward_context():
            held["l"] = fwd()
    ob.run_stage(i, held["l"])
for i in range(len(ob.plan)): stage(i)
print("eager staged max diff", float((opt.flat_g - g_ref).abs().max()))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(len(ob.plan)): stage(i)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graphs = []
for i in range(len(ob.plan)):
    g = torch.cuda.CUDAGraph()
    kw = {"pool": graphs[0].pool()} if (graphs and os.environ.get("DBG_SEPARATE_POOLS") != "1") else {}
    with torch.cuda.graph(g, capture_error_mode="thread_local", **kw):
        stage(i)
    graphs.append(g)
snap = {}
for rep in range(3):
    for gi, g in enumerate(graphs):
        g.replay()
        torch.cuda.synchronize()
        for name, buf in ob.cuts.buffers.items():
            key = (gi, name)
            cur = buf.detach().clone()
            if rep == 0: snap[key] = cur
            elif not torch.equal(snap[key], cur): print("   replay", rep, "after graph", gi, "cut buffer", name, "differs from replay 0: max", float((snap[key]-cur).abs().max()))
        for k, v in held["l"].items():
            key = (gi, "loss:" + k); cur = v.detach().clone()
            if rep == 0: snap[key] = cur
            elif not torch.equal(snap[key], cur): print("   replay", rep, "after graph", gi, "loss", k, "differs")
    torch.cuda.synchronize()
    d = (opt.flat_g - g_ref).abs()
    print("graph staged replay", rep, "max diff", float(d.max()))
    for si, rs in enumerate(ob.ranges):
        for lo, hi in rs:
            print("   stage", si, ob.plan[si]["root"], (lo, hi), "max diff", float(d[lo:hi].max()), "ref max", float(g_ref[lo:hi].abs().max()))
# per-parameter offenders
off = {id(p): o for p, o in zip(opt.params, opt.offsets)}
bad = []
for n, p in model.named_parameters():
    o = off.get(id(p))
    if o is None: continue
    e = float(d[o:o + p.numel()].max())
    if e > 1e-4: bad.append((e, n))
print(sorted(bad, reverse=True)[:15])
