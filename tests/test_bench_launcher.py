"""`python bench.py --gpus N` must itself start N ranks (the driver's SCALE command has no torchrun in front of it) and
rank 0 must print ONE JSON line with n_gpus == N, the per-rank step times and the collectives' share.  Runs the launcher
path of bench.py on CPU + gloo with its stub step (`--workload selftest`); the timed loop, barrier / max-over-ranks
timing and TimedComm accounting are the same code shape as the GPU workloads.  Reference: env-based world discovery in
training/distributed.py:45-108 (the spawn is what torchrun does for the reference)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return p


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_plain_invocation_spawns_n_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout          # rank 0 only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert len(j["per_rank_ms_per_step"]) == 2
    assert j["ms_per_step"] == pytest.approx(max(j["per_rank_ms_per_step"]), rel=1e-6)      # max over ranks
    assert set(j["collective_ms_per_step"]) == {"all_gather", "all_reduce"}
    assert 0.0 < j["collective_share"] <= 1.0


def test_plain_invocation_spawns_8_ranks():
    """The node size of BASELINE.json (8 ranks) through the same launcher path, CPU + gloo."""
    p = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", "selftest"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    (j,) = _json_lines(p.stdout)
    assert j["n_gpus"] == 8 and len(j["per_rank_ms_per_step"]) == 8 and j["config"]["parallelism"] == "dp8"
    assert j["ms_per_step"] == pytest.approx(max(j["per_rank_ms_per_step"]), rel=1e-6)
    assert set(j["collective_ms_per_step"]) == {"all_gather", "all_reduce"}


def test_single_gpu_default_does_not_spawn():
    p = _run(["--steps", "2", "--warmup", "1", "--workload", "selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    (j,) = _json_lines(p.stdout)
    assert j["n_gpus"] == 1 and "per_rank_ms_per_step" not in j


def test_world_size_must_match_gpus():
    # launched as a rank of a 2-rank job but told --gpus 4: refuse instead of printing a wrong n_gpus line
    p = _run(["--gpus", "4", "--workload", "selftest"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_under_torchrun_is_a_rank():
    # the driver's N>1 form: python -m torch.distributed.run ... bench.py --gpus N
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29713", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    (j,) = _json_lines(p.stdout)
    assert j["n_gpus"] == 2


@pytest.mark.parametrize("gpus", ["1", "2"])
def test_result_line_is_the_last_line_of_stdout_even_behind_native_prints(gpus):
    """RCCL prints a version banner through C stdio; on a pipe that buffer is flushed at process exit - AFTER Python's own
    prints (round 6: the five banner lines followed the JSON line of `bench.py --force-dist`).  A driver that reads the last
    line of stdout must still find the result: bench.py flushes C stdio before its line and closes stdout after it.  Here the
    native print is libc `puts` issued before bench's main() in every rank."""
    code = ("import ctypes, sys, runpy; ctypes.CDLL(None).puts(b'native banner line (C stdio, buffered on a pipe)'); "
            f"sys.argv = ['bench.py', '--gpus', '{gpus}', '--steps', '2', '--warmup', '1', '--workload', 'selftest']; "
            f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), lines[-3:]
    assert json.loads(lines[-1])["n_gpus"] == int(gpus)
    assert "native banner line" in p.stdout          # (it was printed - in front of the result, not behind it)
