"""`python bench.py --gpus N` must start its N ranks itself (the driver calls it that way for N = 1; for N > 1 it
wraps it in torch.distributed.run, which must keep working too) and must never silently run fewer ranks.
CPU ranks + gloo + the reduced `--selftest-cpu` shape: only the launcher / barrier / max-over-ranks / JSON plumbing
is under test here (the measured path needs GPUs)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(cmd, extra_env=None, timeout=900):
    env = dict(os.environ, I2P_BENCH_SELFTEST="1", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]


def test_bench_spawns_its_own_ranks():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--selftest-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                           # exactly one line, from rank 0
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "dp2" and line["value"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    # no GPU in the CPU test environment: the measured path must exit non-zero instead of running one rank
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("2 GPUs visible")
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert not _json_lines(r.stdout)


def test_selftest_needs_opt_in():
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--selftest-cpu", "--steps", "1", "--warmup", "0"],
             extra_env={"I2P_BENCH_SELFTEST": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout)
