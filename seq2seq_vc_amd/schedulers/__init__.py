"""Learning-rate schedules (reference schedulers/warmup_lr.py:23-61, noam_lr.py:12-65).

`warmup_lr_value` / `noam_lr_value` are the closed forms; the classes wrap them as torch schedulers for
users who keep a torch optimizer.  With optim.FlatAdam the WarmupLR value is computed on the device inside
the fused optimiser step (csrc/optim.hip), so no scheduler object is needed."""
from torch.optim.lr_scheduler import _LRScheduler


def warmup_lr_value(base_lr, step_num, warmup_steps=4000):
    return base_lr * warmup_steps ** 0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


def noam_lr_value(base_lr, step_num, model_size=320, warmup_steps=25000):
    return base_lr * model_size ** -0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


class WarmupLR(_LRScheduler):
    def __init__(self, optimizer, warmup_steps=4000, last_epoch=-1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, last_epoch)

    def __repr__(self):
        return f"{self.__class__.__name__}(warmup_steps={self.warmup_steps})"

    def get_lr(self):
        return [warmup_lr_value(lr, self.last_epoch + 1, self.warmup_steps) for lr in self.base_lrs]


class NoamLR(_LRScheduler):
    def __init__(self, optimizer, model_size=320, warmup_steps=25000, last_epoch=-1):
        self.model_size, self.warmup_steps = model_size, warmup_steps
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [noam_lr_value(lr, self.last_epoch + 1, self.model_size, self.warmup_steps) for lr in self.base_lrs]


class FusedWarmupLR(object):
    """Scheduler object for optim.FlatAdam.  The WarmupLR value is computed on the device inside the fused optimiser
    step (csrc/optim.hip) from the optimiser's own step counter, so `step()` has nothing to do; the object keeps the
    trainer's `scheduler.step()` / `state_dict()` / `load_state_dict()` call sites valid and its state dict has the keys
    of the reference's `WarmupLR(_LRScheduler)` (schedulers/warmup_lr.py:23-61; `last_epoch`, `_step_count`, `base_lrs`,
    `_last_lr`, `warmup_steps`), so checkpoints written here resume in the reference and the other way round."""

    def __init__(self, optimizer=None, warmup_steps=4000):
        self.optimizer = optimizer
        self.warmup_steps = warmup_steps
        if optimizer is not None and hasattr(optimizer, "warmup_steps"):
            optimizer.warmup_steps = float(warmup_steps)

    def step(self):
        pass

    def state_dict(self):
        k, base = 0, None
        if self.optimizer is not None and hasattr(self.optimizer, "last_stats"):
            k, base = self.optimizer.last_stats()["step"], self.optimizer.lr
        sd = {"warmup_steps": self.warmup_steps, "last_epoch": k, "_step_count": k + 1, "_get_lr_called_within_step": False}
        if base is not None:
            sd["base_lrs"] = [base]
            sd["_last_lr"] = [warmup_lr_value(base, k + 1, self.warmup_steps)]
        return sd

    def load_state_dict(self, sd):
        self.warmup_steps = sd.get("warmup_steps", self.warmup_steps)
        if self.optimizer is not None and hasattr(self.optimizer, "warmup_steps"):
            self.optimizer.warmup_steps = float(self.warmup_steps)
            if sd.get("base_lrs"):
                self.optimizer.lr = float(sd["base_lrs"][0])
