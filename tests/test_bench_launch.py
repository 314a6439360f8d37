"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher in the environment must come back with ONE rank-0 JSON
line that reports N.  CPU: the launch plumbing over gloo (--dry-run: rendezvous + one all-reduce, no GPU
work); the full N-rank run over RCCL is tests/test_parallel_gpu.py (needs >= 2 GPUs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.timeout(300)
def test_bench_self_launches_n_ranks_and_prints_one_line():
    r, lines = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["allreduce_of_ones"] == 2.0 and out["launched_by"] == "self"


@pytest.mark.timeout(120)
def test_bench_single_rank_needs_no_launcher():
    r, lines = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0 and len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
