"""Learning-rate schedules of the training loop (reference: training/scheduler.py:1-64; called once per optimizer step as
`scheduler(step)` in `tri_train_one_epoch`, training/train.py:122-124).  Each factory returns `adjust(step) -> lr` that
also writes the rate into every `optimizer.param_groups[i]["lr"]` - `torch.optim` optimizers and the fused
`vitlens_hip.train.AdamW` (one group) both expose that."""
import math


def assign_learning_rate(optimizer, new_lr):
    for param_group in optimizer.param_groups:
        param_group["lr"] = new_lr


def _warmup_lr(base_lr, warmup_length, step):
    return base_lr * (step + 1) / warmup_length


def _scheduler(optimizer, rate):
    def _lr_adjuster(step):
        lr = rate(step)
        assign_learning_rate(optimizer, lr)
        return lr
    return _lr_adjuster


def const_lr(optimizer, base_lr, warmup_length, steps):
    return _scheduler(optimizer, lambda step: _warmup_lr(base_lr, warmup_length, step) if step < warmup_length else base_lr)


def const_lr_cooldown(optimizer, base_lr, warmup_length, steps, cooldown_steps, cooldown_power=1.0, cooldown_end_lr=0.0):
    start = steps - cooldown_steps

    def rate(step):
        if step < warmup_length:
            return _warmup_lr(base_lr, warmup_length, step)
        if step < start:
            return base_lr
        decay = (1 - (step - start) / (steps - start)) ** cooldown_power          # linear for power 1, polynomial otherwise
        return decay * (base_lr - cooldown_end_lr) + cooldown_end_lr
    return _scheduler(optimizer, rate)


def cosine_lr(optimizer, base_lr, warmup_length, steps):
    def rate(step):
        if step < warmup_length:
            return _warmup_lr(base_lr, warmup_length, step)
        return 0.5 * (1 + math.cos(math.pi * (step - warmup_length) / (steps - warmup_length))) * base_lr
    return _scheduler(optimizer, rate)
