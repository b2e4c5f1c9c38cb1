"""CPU tests of the host-side logic: C-ABI library loads and exports every declared symbol, collaters,
schedules, length bookkeeping, the oracle's optimiser replay, and the 2-rank gloo data-parallel exchange."""
import math
import os
import re
import socket

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_declared_symbols():
    from seq2seq_vc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library(verbose=False)
    L = _lib.lib()
    assert L.s2svc_abi_version() >= 1
    header = open(os.path.join(ROOT, "include", "s2svc_hip.h")).read()
    declared = set(re.findall(r"\b(s2svc_[a-z0-9_]+)\s*\(", header))
    declared -= {"s2svc_operand", "s2svc_gemm_desc"}
    assert declared, "header declares no entry points?"
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/s2svc_hip.h but not exported"
    # and everything the Python binding uses is declared in the header
    for sym in _lib.exported_symbols():
        assert sym in declared or sym in ("s2svc_last_error", "s2svc_abi_version"), f"{sym} missing from the header"


def test_shipped_library_holds_no_packed_fp32_instructions():
    """The library is built without the SLP / loop vectorisers: SLP-formed v_pk_{add,mul,fma}_f32 code gave a wrong high-half
    result in the last 16-lane quarter of a partially active wave, rarely and only inside a full training step (DESIGN.md
    section 5 "Hazard"; tools/repro_spline_slp.py).  Whatever the build flags say, the machine code must hold none of them."""
    from tools import check_no_packed_f32 as chk
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    from seq2seq_vc_amd import _lib
    _lib.lib()                                                     # builds the library if it is missing
    total, per_kernel = chk.count(os.path.join(ROOT, "seq2seq_vc_amd", "csrc", "libs2svc_hip.so"))
    assert not total, f"packed fp32 instructions in the shipped library: {total} in {list(per_kernel)[:5]}"


def test_product_path_fails_loudly_without_gpu():
    from seq2seq_vc_amd.ops import kernels as K
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError):
        K.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-12)


def test_collaters():
    from seq2seq_vc_amd.collaters import ARTTSCollater, ARVCCollater, NARVCCollater
    rng = np.random.default_rng(0)
    batch = [{"src_feat": rng.standard_normal((t, 80)).astype(np.float32), "trg_feat": rng.standard_normal((u, 80)).astype(np.float32),
              "dp_input": rng.standard_normal((t, 80)).astype(np.float32)} for t, u in [(5, 7), (9, 3), (2, 4)]]
    it = ARVCCollater()(batch)
    assert it["xs"].shape == (3, 9, 80) and it["ys"].shape == (3, 7, 80)
    assert it["ilens"].tolist() == [5, 9, 2] and it["olens"].tolist() == [7, 3, 4]
    assert torch.equal(it["xs"][0, 5:], torch.zeros(4, 80)) and it["spembs"] is None
    assert it["labels"][1].tolist() == [0, 0, 1, 1, 1, 1, 1] and it["labels"][0].tolist() == [0] * 6 + [1]
    nar = NARVCCollater()(batch)
    assert set(nar) == {"xs", "ilens", "ys", "olens", "dp_inputs", "dplens", "spembs"} and nar["dplens"].tolist() == [5, 9, 2]
    tts = ARTTSCollater()([(np.array([3, 4, 5]), np.zeros((6, 80), np.float32)), (np.array([7]), np.zeros((2, 80), np.float32))])
    assert tts[0].dtype == torch.long and tts[0].tolist() == [[3, 4, 5], [7, 0, 0]] and tts[3][1].tolist() == [0, 1, 1, 1, 1, 1]


def test_schedulers_match_closed_form():
    from seq2seq_vc_amd.schedulers import WarmupLR, warmup_lr_value
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=8e-5)
    sch = WarmupLR(opt, warmup_steps=4000)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    for k, lr in enumerate(lrs, 1):  # optimiser step k runs with the value for step_num = k
        assert math.isclose(lr, 8e-5 * 4000 ** 0.5 * min(k ** -0.5, k * 4000 ** -1.5), rel_tol=1e-12)
        assert math.isclose(lr, warmup_lr_value(8e-5, k), rel_tol=1e-12)
    assert math.isclose(warmup_lr_value(8e-5, 4000), 8e-5, rel_tol=1e-12)     # peak equals the base lr


def test_subsampled_lengths_follow_the_reference_mask_slicing():
    from seq2seq_vc_amd.modules import Conv2dSubsampling, Lens
    for T in (7, 8, 9, 40, 41, 255, 256):
        t_out = ((T - 1) // 2 - 1) // 2
        for ilen in range(1, T + 1):
            mask = (torch.arange(T) < ilen)[None, None, :]
            ref = int(mask[:, :, :-2:2][:, :, :-2:2].sum())
            assert mask[:, :, :-2:2][:, :, :-2:2].shape[-1] == t_out
            got = Conv2dSubsampling.out_lens(Lens([ilen], "cpu"), t_out).host[0]
            assert got == ref, (T, ilen, got, ref)


def test_oracle_optimizer_replay_matches_torch_adam_with_clipping():
    from oracle import models as OM
    torch.manual_seed(0)
    ps = [torch.randn(7, 5), torch.randn(11)]
    gs = [torch.randn(7, 5) * 3, torch.randn(11) * 3]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=1.0)
    mine = [p.clone() for p in ps]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    for k in range(1, 4):
        lr = OM.warmup_lr(8e-5, k)
        for q, g in zip(ref, gs):
            q.grad = (g * k).clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        OM.adam_step(mine, [g * k for g in gs], state, lr, k)
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b.detach(), atol=1e-7)


def test_logmel_oracle_shapes_and_basis():
    from oracle import logmel as OL
    from seq2seq_vc_amd import frontend as FE
    assert np.abs(FE.mel_basis(16000, 1024, 80, 80, 7600) - OL.mel_filterbank(16000, 1024, 80, 80, 7600)).max() < 1e-7
    x = np.sin(2 * np.pi * 440 * np.arange(8000) / 16000).astype(np.float32) * 0.5
    m = OL.logmelfilterbank(x, 16000, fmin=80, fmax=7600)
    assert m.shape == (1 + 8000 // 256, 80)
    assert int(m.mean(0).argmax()) in range(8, 16)       # 440 Hz lands in the low mel bins
    B = FE.dft_basis(1024)
    fr = np.pad(x, 512, mode="reflect")[:1024]
    ref = np.fft.rfft(fr * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)))
    assert np.abs(B[:513] @ fr - ref.real).max() < 1e-3 and np.abs(B[513:] @ fr - ref.imag).max() < 1e-3


# ---------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seq2seq_vc_amd.distributed import allreduce_mean_, broadcast_
    from oracle import models as OM
    # identical weights everywhere (broadcast), different utterances per rank, mean all-reduce of flat grads
    torch.manual_seed(rank)
    w = torch.randn(1000)
    broadcast_(w, dist, world)
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(8, 1000, generator=g), torch.randn(8, generator=g)
    wr = w.clone().requires_grad_(True)
    ((x @ wr - y) ** 2).mean().backward()
    flat = wr.grad.clone()
    allreduce_mean_(flat, dist, world, chunk_numel=300)        # several chunks
    # bucketed variant of bench.py: two slices reduced independently (second one started before the first is joined),
    # gradients pre-scaled by 1/world through the loss
    from seq2seq_vc_amd.distributed import allreduce_sum_begin, allreduce_end
    flat2 = wr.grad.clone() / world
    h_hi = allreduce_sum_begin(flat2[600:], dist, world, chunk_numel=250)
    h_lo = allreduce_sum_begin(flat2[:600], dist, world, chunk_numel=250)
    allreduce_end(h_hi)
    allreduce_end(h_lo)
    assert torch.allclose(flat2, flat, atol=1e-6), "bucketed async all-reduce != chunked mean all-reduce"
    # numpy arrays travel through the queue by value (torch tensors go through shared-memory handles that die with this process)
    q.put((rank, w.detach().numpy().copy(), wr.grad.numpy().copy(), flat.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_is_the_mean_of_rank_gradients():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, g0, f0), (_, w1, g1, f1) = [(r, *(torch.from_numpy(a) for a in rest)) for r, *rest in out]
    assert torch.equal(w0, w1), "broadcast must make the weights identical"
    assert torch.allclose(f0, (g0 + g1) / 2, atol=1e-6) and torch.equal(f0, f1)


def _rsag_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seq2seq_vc_amd.distributed import _RsAg, allreduce_mean_
    out = []
    for n in (1000, 1003, 7):                               # divisible by the world size, not divisible, smaller than a shard row
        g = torch.Generator().manual_seed(10 * n + rank)
        a = torch.randn(n, generator=g)
        want = a.clone()
        dist.all_reduce(want)                               # the plain sum
        buckets = [_RsAg(a[: n // 2], dist, world), _RsAg(a[n // 2:], dist, world)]      # two buckets in flight, slices of one buffer
        for b_ in buckets:
            b_.finish()
        out.append((n, a.numpy().copy(), want.numpy().copy()))       # by value through the queue (see _dp_worker)
    # chunked buckets (chunk_numel honoured: several collectives per bucket, the last chunk padded)
    g = torch.Generator().manual_seed(999 + rank)
    a = torch.randn(1003, generator=g)
    want = a.clone()
    dist.all_reduce(want)
    b_ = _RsAg(a, dist, world, chunk_numel=250)
    assert len(b_.parts) == 5                                 # 248-element chunks (a multiple of the world size): 4 full + 1 ragged
    b_.finish()
    out.append((-1003, a.numpy().copy(), want.numpy().copy()))
    m = torch.arange(12, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(m, dist, world, chunk_numel=5)
    # torch-optimiser path: a parameter no rank produced a gradient for keeps grad = None (single-GPU semantics: the optimiser skips
    # it); one that only SOME ranks produced gets the mean with zeros for the others
    from seq2seq_vc_amd.distributed import allreduce_grads_
    ps = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(4)),
          torch.nn.Parameter(torch.zeros(5), requires_grad=False)]
    ps[0].grad = torch.full((3,), float(rank + 1))
    if rank == 1:
        ps[2].grad = torch.full((4,), 8.0)
    allreduce_grads_(ps, dist, world)
    assert ps[1].grad is None and ps[3].grad is None, "a gradient nobody produced must stay None"
    assert torch.allclose(ps[0].grad, torch.full((3,), (1 + 2 + 3 + 4) / 4.0)) and torch.allclose(ps[2].grad, torch.full((4,), 2.0))
    q.put((rank, out, m.numpy().copy()))
    dist.destroy_process_group()


def test_four_rank_gloo_reduce_scatter_all_gather_equals_all_reduce():
    """distributed._RsAg (OverlappedBackward(collective="rs_ag")): a bucket reduced as reduce-scatter + all-gather holds the
    all-reduce's sum on every rank -- sizes that the world size divides and that it does not (padded shards), several buckets of
    one buffer in flight; and the chunked mean all-reduce at world size 4."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_rsag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(rank, [(n, torch.from_numpy(g), torch.from_numpy(w_)) for n, g, w_ in out], torch.from_numpy(m)) for rank, out, m in res]
    for rank, out, m in res:
        for n, got, want in out:
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), (rank, n)
            assert torch.equal(got, res[0][1][[k for k, _, _ in res[0][1]].index(n)][1])        # identical on every rank
        assert torch.allclose(m, torch.arange(12, dtype=torch.float32) * 2.5)


def test_gradient_cuts_split_the_backward_pass_without_changing_it():
    """ops.functional.GradCuts (the stage boundaries of distributed.OverlappedBackward): a cut carrying ONE tensor and a cut
    carrying a TUPLE (residual stream + pending feed-forward output of a pre-norm layer stack, one member may be None) give the
    gradients of the uncut graph once every stage has been resumed; above a cut nothing below it has a gradient yet."""
    from seq2seq_vc_amd.ops import functional as Fn

    torch.manual_seed(0)
    w = [torch.randn(6, 6, requires_grad=True) for _ in range(4)]
    x = torch.randn(5, 6)

    def net():
        h = torch.tanh(x @ w[0])
        h = Fn.cut_point(h, "a")
        p = torch.relu(h @ w[1])
        h, p = Fn.cut_point((h, p), "b")
        h = h + p
        h2, none = Fn.cut_point((torch.sigmoid(h @ w[2]), None), "c")
        assert none is None
        return (h2 @ w[3]).pow(2).sum()

    net().backward()
    ref = [t.grad.clone() for t in w]
    for t in w:
        t.grad = None
    cuts = Fn.GradCuts(["a", "b", "c"])
    with Fn.grad_cuts(cuts):
        loss = net()
    loss.backward()
    assert w[3].grad is not None and all(t.grad is None for t in w[:3])          # the backward pass stopped at cut "c"
    cuts.resume("c")
    assert w[2].grad is not None and w[1].grad is None
    cuts.resume("b")
    assert w[1].grad is not None and w[0].grad is None
    cuts.resume("a")
    for got, want in zip([t.grad for t in w], ref):
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    for t in w:
        t.grad = None
    with Fn.grad_cuts(Fn.GradCuts(["b"])):                                       # cut points that are not named stay transparent
        loss = net()
    assert loss.requires_grad
    net().backward()                                                              # and outside the context all of them are
    for got, want in zip([t.grad for t in w], ref):
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_branch_cut_with_shared_leaves_and_the_real_sdp_cut_points():
    """The round-4 stage plans cut INSIDE an auxiliary-stream branch: two conditioning networks feed many consumers (x feeds both
    flow stacks and, added to h_w, the posterior one), the cut carries (x, h_w) as a tuple, the flows' backward pass accumulates in
    the detached leaves and a later stage resumes both networks in ONE autograd call (ops.functional.GradCuts; sdp.py: "sdp_cond" /
    "sdp_cond_x"; distributed.OverlappedBackward with a cut as branch root).  CPU restatement of exactly that shape; and the model
    sources hold the cut points the plans name."""
    from seq2seq_vc_amd.ops import functional as Fn

    torch.manual_seed(1)
    wa, wb = torch.randn(4, 4, requires_grad=True), torch.randn(4, 4, requires_grad=True)      # the two conditioning networks
    wf = [torch.randn(4, 4, requires_grad=True) for _ in range(3)]                              # consumers ("flows")
    inp, w_in = torch.randn(7, 4), torch.randn(7, 4)

    def net(which):
        x, hw = torch.tanh(inp @ wa), torch.tanh(w_in @ wb)
        if which == "both":
            x, hw = Fn.cut_point((x, hw), "sdp_cond")
        elif which == "x":
            x = Fn.cut_point(x, "sdp_cond_x")
        g = x + hw
        post = torch.sigmoid(g @ wf[0]) + torch.sigmoid(g @ wf[1])                               # posterior flows: use x AND h_w
        prior = torch.sigmoid(x @ wf[2]) * post.detach().mean()                                   # prior flows: use x only
        return (post + prior).pow(2).sum()

    ps = [wa, wb] + wf
    net(None).backward()
    ref = [t.grad.clone() for t in ps]
    for which, name, late in (("both", "sdp_cond", (0, 1)), ("x", "sdp_cond_x", (0,))):
        for t in ps:
            t.grad = None
        cuts = Fn.GradCuts([name])
        with Fn.grad_cuts(cuts):
            loss = net(which)
        loss.backward()
        assert all(ps[i].grad is None for i in late), "the networks below the cut must wait for their stage"
        assert all(ps[i].grad is not None for i in range(2, 5))
        cuts.resume(name)
        for got, want in zip([t.grad for t in ps], ref):
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    root = os.path.join(ROOT, "seq2seq_vc_amd")
    sdp_src = open(os.path.join(root, "sdp.py")).read()
    assert 'cut_point((x, h_w), "sdp_cond")' in sdp_src and 'cut_point(x, "sdp_cond_x")' in sdp_src
    assert '"cut:sdp_cond"' in open(os.path.join(root, "models", "aas_vc.py")).read()
    vtn_src = open(os.path.join(root, "models", "vtn.py")).read()
    assert 'cut_point(tuple(head), "decoder_head")' in vtn_src and '"cut:decoder_head"' in vtn_src
    dist_src = open(os.path.join(root, "distributed", "__init__.py")).read()
    assert 'branch_root.startswith("cut:")' in dist_src


def test_lens_bank_keeps_lengths_out_of_a_captured_step():
    """modules.LensBank (captured trainer steps): every Lens made while a bank is active is a slot of one buffer and
    remembers its derivation; refresh() recomputes all slots from new root lengths; max() is the padded length; lengths
    without provenance raise; host tensors tagged with their Lens come back as that Lens."""
    import pytest
    from seq2seq_vc_amd import modules as Mo
    plain = Mo.Lens([3, 5], "cpu")
    assert plain.max() == 5 and plain.cap is None and plain.map(lambda v: v // 2).host == (1, 2) and plain.clamp(4).host == (3, 4)
    bank = Mo.LensBank("cpu")
    src, other = torch.tensor([3, 5]), torch.tensor([3, 5])
    with Mo.lens_bank(bank):
        root = bank.root("ilens", src, src.tolist(), cap=8)
        assert Mo.Lens.of(src, "cpu") is root                      # the very object the trainer registered
        with pytest.raises(RuntimeError):
            Mo.Lens.of(other, "cpu")                               # equal values, no provenance
        with pytest.raises(RuntimeError):
            Mo.Lens([3, 5], "cpu")
        half = root.map(lambda v: v // 2)
        sub = half.map(lambda v: min((v + 3) // 4, 2))
        cl = half.clamp(1)
        assert (half.host, half.cap, half.max()) == ((1, 2), 4, 4) and sub.cap == 1 and cl.host == (1, 1)
        tagged = Mo.tag_lens(torch.tensor(list(half.host)), half)
        assert Mo.Lens.of(tagged, "cpu") is half
    assert Mo.Lens.of(other, "cpu").host == (3, 5)                 # outside the bank: by value, as before
    bank.upload()
    assert bank.buf[:4].tolist() == [[3, 5], [1, 2], [1, 1], [1, 1]]
    assert half.dev.data_ptr() == bank.buf[1].data_ptr()           # the kernels read the slot
    bank.refresh({"ilens": [8, 2]})
    assert bank.buf[:4].tolist() == [[8, 2], [4, 1], [1, 1], [1, 1]] and half.host == (4, 1)
    with pytest.raises(ValueError):
        bank.refresh({"ilens": [9, 2]})                            # longer than the padded length of this graph
    bank.closed = True
    with Mo.lens_bank(bank), pytest.raises(RuntimeError):
        root.map(lambda v: v)


def test_lens_bank_tracks_the_cropped_length_the_reference_computes_on():
    """A banked Lens carries `ext`, the length of the reference's CROPPED tensor (models/vtn.py:208-214: the batch is cut to its
    longest utterance; subsampling.py:74-94: sub(T) = ((T - 1) // 2 - 1) // 2 frames), pushed through the same arithmetic as the
    tensor shapes; crop() is a slot holding B copies of it (the `vlens` of the time-mixing kernels) and follows refresh()."""
    from seq2seq_vc_amd import modules as Mo
    sub = lambda T: ((T - 1) // 2 - 1) // 2          # noqa: E731
    assert Mo.Lens([3, 5], "cpu").crop() is None and Mo.crop_dev(Mo.Lens([3, 5], "cpu")) is None and Mo.crop_dev(None) is None
    bank = Mo.LensBank("cpu")
    src = torch.tensor([100, 57, 81])
    with Mo.lens_bank(bank):
        root = bank.root("ilens", src, src.tolist(), cap=128)
        assert (root.ext, root.cap, root.max()) == (100, 128, 128)
        enc = Mo.Conv2dSubsampling.out_lens(root, sub(128))
        # the reference: mask[:, :, :-2:2][:, :, :-2:2] of a (B, 1, 100) mask has sub(100) = 24 frames; frame t' is valid iff 4 t' < len
        assert enc.host == (24, 15, 21) and enc.ext == 24 and enc.cap == sub(128) == 31
        red = root.map(lambda v: v // 4)
        trim = root.map(lambda v: v - v % 3)
        assert (red.ext, red.cap, trim.ext, trim.cap) == (25, 32, 99, 126)
        c = enc.crop()
        assert c is enc.crop() and c.host == (24, 24, 24) and c.cap == 31 and Mo.crop_dev(enc).data_ptr() == c.dev.data_ptr()
        assert root.crop().host == (100, 100, 100)
    bank.upload()
    assert bank.buf[bank.entries.index((c, ("crop", enc)))].tolist() == [24, 24, 24]
    bank.refresh({"ilens": [60, 128, 9]})          # a batch that fills the padded shape: nothing is absent
    assert enc.host == (15, 31, 3) and enc.ext == 31 and c.host == (31, 31, 31) and root.crop().host == (128,) * 3
    assert (red.ext, trim.ext) == (32, 126)
    bank.refresh({"ilens": [33, 34, 35]})
    assert enc.ext == sub(35) == 8 and enc.host == (8, 8, 8) and c.host == (8, 8, 8)      # ceil(35 / 4) = 9 frames cut to the mask's 8
    k = [i for i, (l, _) in enumerate(bank.entries) if l is c][0]
    assert bank.buf[k].tolist() == [8, 8, 8]
    # without a bank the lengths are what they were: the model has cropped the tensor, t_out is its real length
    plain = Mo.Conv2dSubsampling.out_lens(Mo.Lens([100, 57, 81], "cpu"), sub(100))
    assert plain.host == (24, 15, 21) and plain.ext is None


def test_perm_registry_host_logic():
    """ops.kernels.PermRegistry (state of the step prologue / derived weight copies): the bookkeeping that needs no GPU -- a
    consumer that arrives while a refresh is due runs it first, join() forgets the events."""
    from seq2seq_vc_amd.ops import kernels as K
    reg = K.PermRegistry()
    calls = []

    def refresh():
        reg.due = False
        calls.append(1)

    reg.refresh = refresh
    reg.sync("perm")                       # nothing due, no events: no-op
    assert calls == []
    reg.due = True
    reg.sync()                             # due: refreshed here, once
    reg.sync("perm")
    assert calls == [1] and not reg.due
    reg.join()
    assert reg.ev_perm is None and reg.ev_all is None and reg.waited == set()


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_launches_that_many_ranks_and_refuses_a_mismatch():
    """`python bench.py --gpus N` without a launcher must start N ranks (one process per GPU, env rendezvous on
    127.0.0.1 -- the reference's distributed/launch.py:119-173), n_gpus of the line is what the collective summed over, and a
    WORLD_SIZE that disagrees with --gpus is an error instead of a silent one-rank run.  (--dist-dry-run: gloo, no GPU.)"""
    import json
    r = _run_bench(["--gpus", "2", "--dist-dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"dry_run": True, "n_gpus": 2, "world_size": 2, "ranks_present": 2}
    # under a launcher (WORLD_SIZE set) the flag must agree with it
    r = _run_bench(["--gpus", "2", "--dist-dry-run"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    r = _run_bench(["--gpus", "8", "--dist-dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    # the launcher's env contract, as torch.distributed.run would set it: one rank of a 1-rank job
    r = _run_bench(["--gpus", "1", "--dist-dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                      "MASTER_PORT": str(_free_port_for_tests())})
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout
    # no --gpus under a launcher: the launcher's WORLD_SIZE is the job size (`torchrun --nproc-per-node 8 bench.py`)
    r = _run_bench(["--dist-dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                        "MASTER_PORT": str(_free_port_for_tests())})
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout, r.stderr[-2000:]
    # a rank that dies takes the job down with a non-zero exit code (no hang: the others are terminated)
    r = _run_bench(["--gpus", "2", "--dist-dry-run", "--workload", "nonsense"])
    assert r.returncode != 0


def _free_port_for_tests():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_line_keeps_c3_c5_scalars_where_the_driver_keeps_them():
    """the driver's record keeps scalars inside config / roofline / cpu_baseline and the last 2 000 characters."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"workload": "w"},
           "step_mfma": {"frac_of_bf16_peak": 0.07},
           "roofline": {"bound": "mfma", "frac": 0.4, "avg_launch_us": 100.0,
                        "by_time": {"family": "gemm_dma_kernel", "share_of_kernel_time": 0.3, "mfma_busy": 0.07, "launches_per_step": 140,
                                    "source": "profiles/x", "top5": [{"family": "a"}] * 5}},
           "cpu_baseline": {"value": 5000.0, "cores": 16, "kind": "port"},
           "aasvc": {"ms_per_step": 11.0, "value": 280000.0, "roofline": {"frac": 0.3, "by_time": {"family": "g", "mfma_busy": 0.26}},
                     "step_mfma": {"frac_of_bf16_peak": 0.17}, "cpu_baseline": {"value": 550.0, "cores": 16}, "speedup_vs_cpu_baseline": 509.0},
           "decode": {"value": 3.4e-4, "us_per_step": 350.0, "cpu_baseline": {"value": 0.017}},
           "tts": {"ms_per_step": 4.2, "value": 900000.0, "cpu_baseline": {"value": 3000.0, "cores": 16}, "speedup_vs_cpu_baseline": 300.0},
           "memory_bound": [{"kernel": "k" * 300}] * 6, "alignment": {"mas": {"us_per_utterance": 3.7}}}
    line = bench._shape_line(out)
    cfg, roof, cpu = line["config"], line["roofline"], line["cpu_baseline"]
    assert cfg["aasvc_ms_per_step"] == 11.0 and cfg["decode_rtf"] == 3.4e-4 and cfg["aasvc_cpu_frames_per_s"] == 550.0
    assert cfg["aasvc_roofline_frac"] == 0.3 and cfg["decode_us_per_step"] == 350.0 and cfg["aasvc_speedup_vs_cpu"] == 509.0
    assert roof["by_time_family"] == "gemm_dma_kernel" and roof["by_time_mfma_busy"] == 0.07 and roof["by_time_share"] == 0.3
    assert all(not isinstance(v, (dict, list)) for v in roof.values()), "nested objects inside roofline are dropped by the driver"
    assert cpu["aasvc_value"] == 550.0 and cpu["decode_rtf"] == 0.017
    # C4 (TransformerTTS tts1) is timed too, and what is replayed from profiles/ says so next to the value
    assert cfg["tts_ms_per_step"] == 4.2 and cfg["tts_mel_frames_per_s"] == 900000.0 and cpu["tts_value"] == 3000.0
    assert roof["by_time_source"] == "profiles/x"
    tail = json.dumps(line)[-2000:]
    for key in ("aasvc_ms_per_step", "decode_rtf", "roofline_by_time_mfma_busy", "aasvc_mel_frames_per_s", "cpu_baseline_value", "tts_ms_per_step",
                "tts_mel_frames_per_s"):
        assert f'"{key}"' in tail, key
    keys = list(line)
    assert keys.index("memory_bound") < keys.index("config") < keys.index("roofline") < keys.index("cpu_baseline") < keys.index("decode_rtf")


def test_allreduce_grads_promotes_to_the_widest_dtype():
    """mixed-precision parameter lists (first one bf16) must not round fp32 gradients through bf16."""
    from seq2seq_vc_amd.distributed import allreduce_grads_

    class _FakeDist:
        class ReduceOp:
            SUM = 0

        @staticmethod
        def all_reduce(t, op=None, group=None):
            _FakeDist.seen = t.dtype
            t.mul_(2)           # two ranks holding the same gradients

    a = torch.nn.Parameter(torch.zeros(3, dtype=torch.bfloat16))
    b = torch.nn.Parameter(torch.zeros(4))
    a.grad = torch.ones(3, dtype=torch.bfloat16)
    b.grad = torch.full((4,), 1.0 + 2.0 ** -20)          # not representable in bf16
    allreduce_grads_([a, b], _FakeDist, 2)
    assert _FakeDist.seen == torch.float32
    assert torch.equal(b.grad, torch.full((4,), 1.0 + 2.0 ** -20)) and a.grad.dtype == torch.bfloat16


def test_torch_library_registration_loads_and_lists_every_schema():
    """the kernels are registered with the torch dispatcher (TORCH_LIBRARY in csrc/torch_ops.cpp).  Without a
    GPU: the registration library loads, every op carries the schema string ops/torch_library.py documents, autograd formulas are
    attached, and a call on CPU tensors is refused by the dispatcher (no CPU implementation exists -- no fallback)."""
    import torch
    from seq2seq_vc_amd import _lib
    from seq2seq_vc_amd.ops import torch_library as TL
    _lib.build_torch_ops(verbose=False)
    ops = TL.load()
    for name, schema in TL.SCHEMAS.items():
        assert str(getattr(ops, name).default._schema) == schema, name
    assert ops.abi_version() == _lib.lib().s2svc_abi_version()
    with pytest.raises(NotImplementedError):
        ops.mas_forward(torch.zeros(1, 4, 3), torch.tensor([3]), torch.tensor([4]))
    with pytest.raises(NotImplementedError):
        ops.gemm_bias_act(torch.zeros(4, 8), torch.zeros(2, 8), None, "relu")
    # every *_bwd op has its forward op, and the differentiable forward ops have an autograd kernel registered
    for fwd in ("ctc_forward_sum", "masked_l1_bce", "guided_attn_loss", "attn_fwd", "ln_residual_dropout"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"s2svc::{fwd}", "Autograd"), fwd


def test_committed_step_timelines_hold_no_aten_kernels():
    """a captured training step launches no ATen kernel -- checked on the kernel-name lists of the committed rocprofv3
    timelines of this round (tools/rocpd_timeline.py over `bench.py` under rocprofv3 --kernel-trace).  The one exception is the draw of the
    stochastic duration predictor (duration_predictor.py:247-254: torch.randn): its normal_ kernel and the two int64 fills with which
    torch's graph-safe generator hands seed and offset to a replay."""
    prof = os.path.join(ROOT, "profiles")
    allowed = ("distribution_elementwise_grid_stride_kernel", "FillFunctorIl")
    seen = {}
    for wl, budget in (("vtn", 0), ("tts", 0), ("aasvc", 3)):
        path = os.path.join(prof, f"r06_{wl}_train_bf16_timeline.txt")
        assert os.path.exists(path), path
        lines = [ln for ln in open(path) if "at6native" in ln or "at::native" in ln]
        bad = [ln.strip()[-120:] for ln in lines if not any(a in ln for a in allowed)]
        assert not bad, f"{wl}: ATen kernels in the captured step: {bad[:5]}"
        assert len(lines) <= budget, f"{wl}: {len(lines)} ATen launches (allowed {budget})"
        seen[wl] = len(lines)
        n = [ln for ln in open(path) if ln.startswith("# step:")]
        assert n, "timeline header missing"
    assert seen == {"vtn": 0, "tts": 0, "aasvc": seen["aasvc"]}
