"""Trainers resolved by name (`trainer_type:`), mirroring `seq2seq_vc.trainers`
(reference trainers/base.py:18-227, ar_vc.py:28-112, ar_tts.py:22-100, aas_vc.py:22-164).

Same constructor `(steps, epochs, data_loader, sampler, model, vocoder, criterion, optimizer, scheduler,
config, device)` and `run / save_checkpoint / load_checkpoint / load_trained_modules / freeze_modules`;
`_train_step` reproduces the reference's loss composition, gradient-accumulation scaling, clip -> step ->
scheduler order and step counting.  Plotting / vocoding of intermediate results (matplotlib, soundfile,
tensorboardX) is outside the hot path and not reproduced; a log callback receives the averaged losses.

Host synchronisation: the reference calls `.item()` 4-6 times per step; here losses are accumulated in a
device tensor and read once per `log_interval_steps`.
"""
import contextlib
import logging
import os
from collections import OrderedDict, defaultdict

import torch

from .. import distributed as D
from ..optim import FlatAdam
from ..ops import functional as Fn
from ..ops import kernels as K


def get_partial_state_dict(model_state_dict, modules):
    return OrderedDict((k, v) for k, v in model_state_dict.items() if any(k.startswith(m) for m in modules))


def freeze_modules(model, modules):
    """requires_grad=False for every parameter whose name starts with one of `modules` (utils/model_io.py:95-111)."""
    for name, p in model.named_parameters():
        if any(name.startswith(m) for m in modules):
            logging.warning(f"Freezing {name}. It will not be updated during training.")
            p.requires_grad = False
    return model, filter(lambda x: x.requires_grad, model.parameters())


class Trainer(object):
    def __init__(self, steps, epochs, data_loader, sampler, model, vocoder, criterion, optimizer, scheduler, config,
                 device=torch.device("cpu")):
        self.steps, self.epochs = steps, epochs
        self.data_loader, self.sampler = data_loader, sampler
        self.model, self.vocoder = model, vocoder
        self.criterion, self.optimizer, self.scheduler = criterion, optimizer, scheduler
        self.config, self.device = config, torch.device(device)
        self.finish_train = False
        self.gradient_accumulate_steps = self.config.get("gradient_accumulate_steps", 1)
        self.backward_steps = 0
        self.loss_names = []
        self.loss_acc = None            # device tensor: running sums of the logged losses
        self.total_train_loss = defaultdict(float)
        self.log_fn = None              # optional callable(steps, dict) -- stands in for tensorboardX
        self._schedule_gradient_work()
        self._setup_data_parallel()
        self._capture = None            # set while a staged step is being captured (trainers/graphed.py)
        self._graphed = None
        # Captured steps (trainers/graphed.py) are the DEFAULT wherever they apply (config["hip_graph"] absent or "auto"): they compute
        # what the eager step computes -- the reference's step on the batch cropped to its longest utterance, GPU case
        # captured_steps_on_short_batches_vs_oracle -- 3-4 x faster (the eager step is bound by the host).  True / "trace" demand
        # them (and raise where they do not apply), False turns them off.
        mode = self.config.get("hip_graph", "auto")
        if mode and self.device.type == "cuda":
            why = self._captured_steps_unavailable()
            if mode == "auto":
                if why is None:
                    from .graphed import GraphedStep
                    self._graphed = GraphedStep(self)
                else:
                    logging.info(f"hip_graph: eager steps ({why})")
            else:
                if not isinstance(self.optimizer, FlatAdam):
                    raise ValueError('config["hip_graph"] needs the fused optimiser (optim.FlatAdam)')
                from .graphed import GraphedStep
                self._graphed = GraphedStep(self)

    # Captured steps (trainers/graphed.py): tensor field of the batch -> (its length field, padding value), None = no captured
    # step for this trainer; _graph_regime(): whatever host-side state changes WHAT a step launches (one set of graphs each).
    GRAPH_BATCH = None
    GRAPH_ACCUMULATE = False        # True: the trainer's _graph_regime() names the micro-step's role in an accumulation window

    def _graph_regime(self):
        return ()

    def _captured_steps_unavailable(self):
        """None if this trainer can replay its steps from hipGraphs, else the reason (config["hip_graph"] = "auto")."""
        from ..schedulers import FusedWarmupLR
        if not isinstance(self.optimizer, FlatAdam):
            return "the optimiser is not optim.FlatAdam"
        if self.GRAPH_BATCH is None:
            return f"{type(self).__name__} has no captured step"
        if self.gradient_accumulate_steps != 1 and not self.GRAPH_ACCUMULATE:
            return "gradient accumulation without captured micro-steps"
        if self.scheduler is not None and not isinstance(self.scheduler, FusedWarmupLR):
            return "a host-side learning-rate scheduler"
        if self.dist is not None and self.dp is None and getattr(self, "fx", None) is None:
            return "data parallelism without a staged backward pass (model.dp_plan)"
        return None

    def _batch_dict(self, batch):
        if not isinstance(batch, dict):
            raise NotImplementedError(f"{type(self).__name__}: captured steps take dict batches")
        return batch

    def _step(self, batch):
        if self._graphed is not None:
            self._graphed.step(batch)
        else:
            self._train_step(batch)

    # how parameter-gradient kernels are scheduled (ops/functional.py, "Side streams"): (side streams, inline batches)
    GRADIENT_WORK = (4, False)
    GRADIENT_BATCH = None           # closures per gradient batch (None: ops.functional.enable_side_streams' default, 16 forked / 64 inline)
    DP_GRAD_PAYLOAD = "fp32"        # dtype of the data-parallel gradient exchange (config["dp_grad_payload"] overrides)

    def _schedule_gradient_work(self):
        if self.device.type == "cuda" and isinstance(self.optimizer, FlatAdam):
            n, inline = self.GRADIENT_WORK
            Fn.enable_side_streams(self.config.get("side_streams", n), inline_batches=self.config.get("inline_batches", inline),
                                   batch=self.config.get("gradient_batch", self.GRADIENT_BATCH))

    # -- data parallelism (reference: apex DistributedDataParallel wrap, bin/vc_train.py:423-431) ------------------------
    def _setup_data_parallel(self):
        """config["distributed"] (set by the launcher code as in bin/vc_train.py:197-201 after init_process_group): every rank
        starts from rank 0's parameters and buffers, and every optimiser step averages the gradients over the ranks -- with
        FlatAdam stage by stage, overlapped with the backward pass (distributed.OverlappedBackward); with a torch optimiser
        one coalesced all-reduce of p.grad after backward.  BatchNorm statistics stay rank-local, rank 0 checkpoints."""
        self.dist, self.world, self.dp, self.fx = None, 1, None, None
        if not self.config.get("distributed", False):
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('config["distributed"] is set but torch.distributed is not initialised '
                               "(call init_process_group / distributed.init_from_env first, as bin/vc_train.py:197-201 does)")
        self.dist, self.world = dist, dist.get_world_size()
        self.config.setdefault("rank", dist.get_rank())
        D.broadcast_model_(self._net(), self.optimizer, dist, self.world)
        exchange = self.config.get("dp_exchange", "flush" if self.config.get("dp_collective", "allreduce") == "allreduce" else "stages")
        if isinstance(self.optimizer, FlatAdam) and exchange == "flush" and self.optimizer.flat_g.is_cuda:
            # round 6: the UNCUT backward pass, every bucket exchanged behind the flush of the gradient batch that finished it
            # (distributed.FlushExchange): no joins inside the backward pass, buckets at the granularity of the gradient batches
            self.fx = D.FlushExchange(self.optimizer, dist, self.world, payload=self.config.get("dp_grad_payload", self.DP_GRAD_PAYLOAD),
                                      min_bucket_numel=int(self.config.get("dp_min_bucket_mb", 16.0) * 262144))
        elif isinstance(self.optimizer, FlatAdam) and hasattr(self._net(), "dp_plan"):
            self.dp = D.OverlappedBackward(self._net(), self.optimizer, dist, self.world,
                                           payload=self.config.get("dp_grad_payload", self.DP_GRAD_PAYLOAD),
                                           collective=self.config.get("dp_collective", "allreduce"))

    def _forward_context(self):
        """Context of the forward pass: activates the model's gradient cuts when the backward pass runs in stages."""
        return self.dp.forward_context() if self.dp is not None else contextlib.nullcontext()

    def _begin_step(self, zero=True):
        """First call of a training step: clear the gradients (zero = None: only if a deferred zero_grad is pending) and, with
        the fused optimiser, start the step prologue beside the forward pass (optim.FlatAdam.begin_step)."""
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.begin_step(zero=zero)
        elif zero:
            self.optimizer.zero_grad()

    def _backward(self, total, parts, last_micro_step=True):
        """Backward pass + (on the last micro-step of an accumulation window) the gradient exchange.
        total: the scalar loss (already divided by gradient_accumulate_steps); parts: the same loss split by the keys of
        model.dp_plan() (sum(parts) == total), used when the backward pass runs stage by stage."""
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.join_prologue()       # zero-fill of the gradients + refreshed weight copies: done before the first gradient
        if self.dp is not None:
            if self._capture is not None and last_micro_step:   # being captured stage by stage: the exchange runs between the replays
                self._capture.staged_backward(self.dp, parts)
            else:
                self.dp.backward(parts, reduce=last_micro_step)
        elif self.fx is not None:
            fx = self.fx
            scale = 1.0 / self.world             # SUM exchange of gradients of loss / world == the mean the reference's DDP takes
            if not last_micro_step:              # an accumulation micro-step: no exchange
                Fn.root_backward(total, scale)
                Fn.side_join()
            elif not fx.select(self._graph_regime()):      # first step of this regime (eager): learn the plan while it runs
                fx.learn_begin()
                try:
                    Fn.root_backward(total, scale)
                    Fn.side_join()
                finally:
                    fx.learn_end()
                for h in D.allreduce_sum_begin(self.optimizer.flat_g, self.dist, self.world):
                    h.wait()
            elif self._capture is not None:      # being captured: marks become nodes of the graph, the exchange follows its launch
                self._capture.flush_backward(fx, total, scale)
            else:
                with fx.recording():
                    Fn.root_backward(total, scale)
                    Fn.side_join()
                fx.mark_end()
                fx.issue()
                fx.finish()
        else:
            Fn.root_backward(total)
            Fn.side_join()
            if self.dist is not None and last_micro_step:
                if isinstance(self.optimizer, FlatAdam):
                    D.allreduce_mean_(self.optimizer.flat_g, self.dist, self.world)
                else:
                    D.allreduce_grads_(list(self._net().parameters()), self.dist, self.world)

    # -- core loop -------------------------------------------------------------------------------
    def _net(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def run(self):
        self.backward_steps = 0
        while not self.finish_train:
            self._train_epoch()
        logging.info("Finished training.")

    def _train_epoch(self):
        n = 0
        for n, batch in enumerate(self.data_loader["train"], 1):
            self._step(batch)
            if self.backward_steps % self.gradient_accumulate_steps > 0:
                continue
            if self.config.get("rank", 0) == 0:
                self._check_log_interval()
                self._check_save_interval()
            if self.finish_train:
                return
        self.epochs += 1
        self.train_steps_per_epoch = n
        if self.config.get("distributed", False) and self.sampler and self.sampler.get("train") is not None:
            self.sampler["train"].set_epoch(self.epochs)

    def _accumulate(self, **losses):
        """Sum losses on the device (no host sync); names are fixed by the first call."""
        if self.loss_acc is None:
            self.loss_names = list(losses)
            self.loss_acc = torch.zeros(len(self.loss_names), dtype=torch.float32, device=self.device)
        if self.device.type == "cuda" and len(self.loss_names) <= 8:
            # one launch of this library (csrc/glue.hip): acc[i] += sum(loss_i) / accumulation steps -- a term may be a vector (the
            # per-utterance duration NLL), which is summed on the way
            terms = []
            for k in self.loss_names:
                v = losses[k]
                v = v.detach() if isinstance(v, torch.Tensor) else Fn.const_scalar(self.device, float(v))
                if v.dtype != torch.float32 or not v.is_contiguous() or v.device != self.device:
                    v = v.to(device=self.device, dtype=torch.float32).contiguous()
                terms.append((v, 1.0 / self.gradient_accumulate_steps))
            K.scalars_axpy(terms, self.loss_acc, beta=1.0)
            return
        vals = torch.stack([torch.as_tensor(losses[k], dtype=torch.float32, device=self.device).detach().sum().reshape(())
                            for k in self.loss_names])
        self.loss_acc += vals / self.gradient_accumulate_steps

    def _optimizer_step(self):
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.step()               # clip + WarmupLR + Adam fused on the device
        else:
            if self.config["grad_norm"] > 0:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.config["grad_norm"])
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

    def _check_log_interval(self):
        if self.steps % self.config["log_interval_steps"] == 0 and self.loss_acc is not None:
            vals = (self.loss_acc / self.config["log_interval_steps"]).tolist()   # the one host sync
            self.total_train_loss = dict(zip(self.loss_names, vals))
            for k, v in self.total_train_loss.items():
                logging.info(f"(Steps: {self.steps}) {k} = {v:.4f}.")
            if self.log_fn is not None:
                self.log_fn(self.steps, dict(self.total_train_loss))
            self.loss_acc.zero_()

    def _check_save_interval(self):
        if self.steps % self.config["save_interval_steps"] == 0:
            self.save_checkpoint(os.path.join(self.config["outdir"], f"checkpoint-{self.steps}steps.pkl"))
            logging.info(f"Successfully saved checkpoint @ {self.steps} steps.")

    def _check_train_finish(self):
        if self.steps >= self.config["train_max_steps"]:
            self.finish_train = True

    # -- checkpoints (trainers/base.py:85-124: {model, optimizer, scheduler, steps, epochs}) --------
    def save_checkpoint(self, checkpoint_path):
        sd = {"optimizer": self.optimizer.state_dict(),
              "scheduler": self.scheduler.state_dict() if self.scheduler is not None else {},
              "steps": self.steps, "epochs": self.epochs,
              "model": OrderedDict((k, v.detach().cpu().clone()) for k, v in self._net().state_dict().items())}
        os.makedirs(os.path.dirname(checkpoint_path) or ".", exist_ok=True)
        torch.save(sd, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, load_only_params=False):
        sd = torch.load(checkpoint_path, map_location="cpu")
        self._net().load_state_dict(sd["model"])
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.refresh_shadow()
        if not load_only_params:
            self.steps, self.epochs = sd["steps"], sd["epochs"]
            self.optimizer.load_state_dict(sd["optimizer"])
            if self.scheduler is not None:
                self.scheduler.load_state_dict(sd["scheduler"])
        if self.dist is not None:
            D.broadcast_model_(self._net(), self.optimizer, self.dist, self.world)

    def load_trained_modules(self, checkpoint_path, init_mods):
        """Partial (prefix-filtered, shape-verified) load: trainers/ar_vc.py:31-57 + utils/model_io.py:12-92."""
        main = self._net().state_dict()
        src = torch.load(checkpoint_path, map_location="cpu")["model"]
        missing = [m for m in init_mods if not any(k.startswith(m) for k in src)]
        if missing:
            raise ValueError(f"Specified module(s) don't match the pre-trained model: {missing}")
        part = get_partial_state_dict(src, init_mods)
        want = sorted((k, tuple(v.shape)) for k, v in main.items() if any(k.startswith(m) for m in init_mods))
        have = sorted((k, tuple(v.shape)) for k, v in part.items())
        if want != have:
            raise ValueError(f"modules do not match: pre-trained-only {set(have) - set(want)}, model-only {set(want) - set(have)}")
        for k in part:
            logging.warning(f"Overriding module {k}")
        main.update(part)
        self._net().load_state_dict(main)
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.refresh_shadow()
        if self.dist is not None:
            D.broadcast_model_(self._net(), self.optimizer, self.dist, self.world)

    def freeze_modules(self, modules):
        freeze_modules(self.model, modules)
        if self._graphed is not None:        # what a step launches has changed: captured graphs of the old set are dropped
            self._graphed.invalidate()


class ARVCTrainer(Trainer):
    """trainers/ar_vc.py:59-112: loss = l1 + bce (+ guided attention); zero_grad BEFORE backward."""

    GRAPH_BATCH = {"xs": ("ilens", 0.0), "ys": ("olens", 0.0), "labels": ("olens", 1.0)}

    def _forward_losses(self, batch):
        dev = self.device
        xs, ys, labels = batch["xs"].to(dev), batch["ys"].to(dev), batch["labels"].to(dev)
        ilens, olens = batch["ilens"], batch["olens"]       # stay on the host (sizes only)
        after, before, logits, ys_, labels_, olens_, (att_ws, ilens_ds_st, olens_in) = self.model(xs, ilens, ys, labels, olens)
        l1, bce = self.criterion["Seq2SeqLoss"](after, before, logits, ys_, labels_, olens_)
        terms = [(l1, 1.0), (bce, 1.0)]
        logs = {"train/l1_loss": l1, "train/bce_loss": bce}
        if self.config.get("use_guided_attn_loss", False):
            ga = self.criterion["guided_attn"](self._guided_attention_input(att_ws), ilens_ds_st, olens_in)
            terms.append((ga, 1.0))
            logs["train/guided_attn_loss"] = ga
        loss = Fn.weighted_sum(terms)            # l1 + bce (+ guided attention): one launch each way (ar_vc.py:86-97)
        logs["train/loss"] = loss
        return loss, logs

    def _guided_attention_input(self, att_ws):
        """The reference hands `att_ws` -- VTN returns a LIST of per-layer (B, H, T_out, T_in) maps, last layer first
        (models/vtn.py:276-290; the torch.cat is commented out there) -- straight to the loss, which would fail on a list;
        no recipe enables it.  Here the configured intent is built the way ESPnet (and the reference's own
        TransformerTTS.forward, models/transformer_tts.py:205-222) does: the last `num_layers_applied_guided_attn` layers,
        the first `num_heads_applied_guided_attn` heads of each, concatenated along the head axis."""
        if isinstance(att_ws, torch.Tensor):
            return att_ws
        net = self._net()
        n_layers = getattr(net, "num_layers_applied_guided_attn", 2)
        n_heads = getattr(net, "num_heads_applied_guided_attn", 2)
        return torch.cat([a[:, :n_heads] for a in att_ws[:n_layers]], dim=1)

    def _train_step(self, batch):
        K.reset_op_counter()
        K.advance_seed(self.device)
        self._begin_step()                      # zero_grad (the reference clears before backward: nothing reads them in between)
        with self._forward_context():
            loss, logs = self._forward_losses(batch)
        self._accumulate(**logs)
        self._backward(loss, {"loss": loss})
        self.backward_steps += 1
        self._optimizer_step()
        self.steps += 1
        self._check_train_finish()


class ARTTSTrainer(ARVCTrainer):
    """trainers/ar_tts.py:45-100: identical composition; the TTS collater yields a tuple."""

    def _batch_dict(self, batch):
        if not isinstance(batch, dict):
            xs, ilens, ys, labels, olens = batch[:5]
            batch = {"xs": xs, "ilens": ilens, "ys": ys, "labels": labels, "olens": olens}
        return batch

    def _forward_losses(self, batch):
        return super()._forward_losses(self._batch_dict(batch))


class AASVCTrainer(Trainer):
    """trainers/aas_vc.py:56-164: l1 + lambda_align*(forward_sum + bin) + sum(dur_nll) [after dp_train_start_steps];
    gradient accumulation divides the loss; zero_grad AFTER the optimiser step."""

    GRADIENT_WORK = (0, True)       # chip-filling kernels: batched on the issuing stream, not forked (17.6 vs 20.9 ms/step)
    DP_GRAD_PAYLOAD = "fp32"        # the reference's DDP all-reduces fp32 gradients: the parity setting is the default.  630 MB of fp32
    #                                 gradients per step (vc2) are 7 ms on one xGMI link against a 12 ms step: a multi-GPU recipe opts
    #                                 into config["dp_grad_payload"] = "bf16" (the exchange runs on a bf16 copy, ~3 significant digits
    #                                 in the 8-rank sum; multi-GPU runs then no longer match single-GPU runs bit for bit)
    GRAPH_BATCH = {"xs": ("ilens", 0.0), "ys": ("olens", 0.0), "dp_inputs": ("dplens", 0.0)}

    GRAPH_ACCUMULATE = True         # micro-steps of an accumulation window are captured by role (trainers/graphed.py)

    def _graph_regime(self):
        """What the captured step depends on besides shapes: whether the duration loss is on, and the ROLE of the coming micro-step
        in its accumulation window -- a deferred zero-fill is due at its start / the optimiser step follows its backward pass."""
        last = (self.backward_steps + 1) % self.gradient_accumulate_steps == 0
        zero_due = bool(getattr(self.optimizer, "_zero_due", False))
        return (self.steps > self.config.get("dp_train_start_steps", 0), zero_due, last)

    def _train_step(self, batch):
        dev = self.device
        K.reset_op_counter()
        K.advance_seed(dev)
        self._begin_step(zero=None)             # the zero_grad deferred by the previous optimiser step runs beside this forward pass
        net = self._net()
        if getattr(net, "forward_sum_prefetch", 0) is None and "ForwardSumLoss" in self.criterion:
            net.forward_sum_prefetch = self.criterion["ForwardSumLoss"].prefetch     # its recursion runs beside the decoder
        xs, ys, dp_inputs = batch["xs"].to(dev), batch["ys"].to(dev), batch["dp_inputs"].to(dev)
        with self._forward_context():
            ret = self.model(xs, batch["ilens"], ys, batch["olens"], dp_inputs, dp_lengths=batch["dplens"])
            # loss = l1 + lambda_align * (forward_sum + bin) + duration (aas_vc.py:100-139): the sums -- and the division by the
            # accumulation steps (:141-143) -- are weights of ONE launch each way (Fn.weighted_sum), not a chain of 0-dim ATen ops
            zero = Fn.const_scalar(dev, 0.0)
            lam, gas = float(self.config["lambda_align"]), float(self.gradient_accumulate_steps)
            logs = {}
            dec_terms = []
            if "L1Loss" in self.config["criterions"]:
                l1 = self.criterion["L1Loss"](ret["after_outs"], ret["before_outs"], ret["ys"], ret["olens"])
                logs["train/l1_loss"] = l1
                dec_terms.append((l1, 1.0))
            fs = self.criterion["ForwardSumLoss"](ret["log_p_attn"], ret["ilens"], ret["olens_reduced"])
            logs["train/forward_sum_loss"], logs["train/binary_loss"] = fs, ret["bin_loss"]
            align_terms = [(fs, lam), (ret["bin_loss"], lam)]
            dur = zero
            if self.steps > self.config.get("dp_train_start_steps", 0):
                if "DurationPredictorLoss" in self.config["criterions"]:
                    dur = self.criterion["DurationPredictorLoss"](ret["d_outs"], ret["ds"], ret["ilens"])
                    align_terms.append((dur, 1.0))
                elif "StochasticDurationPredictorLoss" in self.config["criterions"]:
                    dur = ret["dur_nll"]                       # (B,): summed inside the weighted sum / the logging launch
                    align_terms.append((dur, 1.0))
            logs["train/duration_loss"] = dur
            if self.dp is not None:                            # staged backward pass: one root per key of model.dp_plan()
                dec_loss = Fn.weighted_sum([(t, w / gas) for t, w in dec_terms]) if dec_terms else zero
                align_loss = Fn.weighted_sum([(t, w / gas) for t, w in align_terms])
                parts = {"decoder": dec_loss, "align": align_loss}
                loss = None
                logs["train/loss"] = Fn.weighted_sum([(t.detach(), w) for t, w in dec_terms + align_terms])
            else:
                parts = None
                loss = Fn.weighted_sum([(t, w / gas) for t, w in dec_terms + align_terms])
                logs["train/loss"] = loss if gas == 1.0 else Fn.weighted_sum([(t.detach(), w) for t, w in dec_terms + align_terms])
        self._accumulate(**logs)
        self.backward_steps += 1
        last = self.backward_steps % self.gradient_accumulate_steps == 0
        self._backward(loss, parts, last_micro_step=last)
        if not last:
            return
        self._optimizer_step()
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.zero_grad(defer=True)
        else:
            self.optimizer.zero_grad()
        self.steps += 1
        self._check_train_finish()


class NARVCTrainer(Trainer):
    """trainers/nar_vc.py:53-103 (FastSpeechVC with teacher durations): loss = l1 + duration MSE (log domain); zero_grad
    BEFORE backward; clip -> step -> scheduler."""

    GRADIENT_WORK = (0, True)

    def _train_step(self, batch):
        dev = self.device
        K.reset_op_counter()
        K.advance_seed(dev)
        xs, ys, dp_inputs = batch["xs"].to(dev), batch["ys"].to(dev), batch["dp_inputs"].to(dev)
        self._begin_step()
        with self._forward_context():
            before, after, d_outs, ilens_, olens_, ys_ = self.model(xs, batch["ilens"], ys, batch["olens"], batch["durations"],
                                                                   batch["duration_lens"], dp_inputs, dp_lengths=batch["dplens"])
            l1 = self.criterion["L1Loss"](after, before, ys_, olens_)
            dur = self.criterion["DurationPredictorLoss"](d_outs, batch["durations"].to(dev), ilens_)
            loss = l1 + dur
        self._accumulate(**{"train/l1_loss": l1, "train/duration_loss": dur, "train/loss": loss})
        self._backward(loss, {"decoder": l1, "duration": dur})
        self.backward_steps += 1
        self._optimizer_step()
        self.steps += 1
        self._check_train_finish()
