"""CPU: the learning-rate schedules (training/scheduler.py) against the imported reference (build container only) and on
the fused AdamW's `param_groups` (what `scheduler(step)` writes is what the next `optimizer_step` uses)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cosine_lr", dict(base_lr=5e-4, warmup_length=10, steps=100)), ("const_lr", dict(base_lr=2e-4, warmup_length=5, steps=50)),
         ("const_lr_cooldown", dict(base_lr=1e-3, warmup_length=8, steps=60, cooldown_steps=20, cooldown_power=2.0, cooldown_end_lr=1e-5)),
         ("const_lr_cooldown", dict(base_lr=1e-3, warmup_length=0, steps=40, cooldown_steps=10))]

_REF = r'''
import json, sys
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
import training.scheduler as S
class Opt:
    def __init__(self): self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]
out = []
for name, kw in json.loads(sys.argv[2]):
    o = Opt(); f = getattr(S, name)(o, **kw)
    out.append([[float(f(s)), o.param_groups[0]["lr"], o.param_groups[1]["lr"]] for s in range(kw["steps"])])
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_schedules_equal_the_reference():
    from training import scheduler as S
    r = subprocess.run([sys.executable, "-c", _REF, os.path.join(ROOT, "oracle"), json.dumps(CASES)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]
    for (name, kw), want in zip(CASES, ref):
        o = Opt(); f = getattr(S, name)(o, **kw)
        for s in range(kw["steps"]):
            lr = f(s)
            assert abs(lr - want[s][0]) <= 1e-18 + 1e-15 * abs(want[s][0]), (name, s, lr, want[s][0])
            assert o.param_groups[0]["lr"] == lr and o.param_groups[1]["lr"] == lr


def test_schedule_drives_the_fused_adamw():
    from training.scheduler import cosine_lr
    from vitlens_hip.train import AdamW
    opt = AdamW({"w": torch.zeros(4, 4)}, lr=5e-4)
    assert opt.lr == 5e-4 and opt.param_groups[0]["lr"] == 5e-4
    sch = cosine_lr(opt, 5e-4, warmup_length=4, steps=20)
    assert sch(0) == 5e-4 / 4 and opt.lr == 5e-4 / 4
    sch(4)
    assert opt.lr == 5e-4
    sch(19)
    assert 0 < opt.lr < 5e-4 * 0.01 and isinstance(opt.lr, float)
    opt.lr = 1e-3
    assert opt.param_groups[0]["lr"] == 1e-3
