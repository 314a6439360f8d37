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


@pytest.mark.timeout(300)
def test_bench_scaling_flags_reach_every_rank():
    """the flags of the first multi-GPU run (tools/scale_run.sh): strong scaling splits --global-batch over the ranks, --linear-schedule and
    --bucket-dtype are accepted and echoed by the launch plumbing"""
    r, lines = _run(["--gpus", "2", "--dry-run", "--global-batch", "256", "--linear-schedule", "persistent", "--bucket-dtype", "bf16"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["flags"] == {"linear_schedule": "persistent", "bucket_dtype": "bf16", "global_batch": 256, "per_gpu_batch": 128, "scaling": "strong"}
    r, lines = _run(["--gpus", "2", "--dry-run"])
    assert json.loads(lines[0])["flags"] == {"linear_schedule": "auto", "bucket_dtype": "f32", "global_batch": 0, "per_gpu_batch": 32, "scaling": "weak"}
    r, _ = _run(["--gpus", "1", "--dry-run", "--bucket-dtype", "fp8"])
    assert r.returncode != 0


def test_rccl_summary_reads_the_debug_lines(tmp_path):
    """`collective.rccl` quotes RCCL's own description of the communicator: parsed from NCCL_DEBUG=INFO text (format of RCCL 2.2x)"""
    sys.path.insert(0, ROOT)
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "host:1:1 [0] NCCL INFO RCCL version 2.26.6+hip7.0 HEAD:abc",
        "host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 1000 - Init START",
        "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC",
        "host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 7[7] via P2P/IPC",
        "host:1:1 [0] NCCL INFO Channel 01/0 : 3[3] -> 4[4] via P2P/IPC",
        "host:1:1 [0] NCCL INFO 16 coll channels, 16 nvls channels, 16 p2p channels, 2 p2p channels per peer"]))
    s = bench.rccl_summary(str(log))
    assert s["channels"] == 2 and s["transports"] == ["P2P/IPC"] and s["peers_of_rank0"] == [1, 7]
    assert s["transport_of_rank0"] == "xGMI peer-to-peer" and any("RCCL version" in ln for ln in s["lines"])
    assert "not collected" in bench.rccl_summary(None)["debug_lines"]


def test_power_sampler_reads_hwmon_and_picks_the_loaded_card(tmp_path):
    """bench.py's `power` object: board power / cap / shader clock from the amdgpu hwmon files, the card under load = the one with the highest
    mean power (the container sees every card of the host), joules per step = mean power x step time; no card readable -> an error entry, not
    an exception"""
    import time
    sys.path.insert(0, ROOT)
    import bench
    for i, (uw, cap, hz) in enumerate(((250_000_000, 1_400_000_000, 97_000_000), (1_395_000_000, 1_400_000_000, 2_135_000_000))):
        d = tmp_path / f"class/drm/card{i}/device/hwmon/hwmon{i}"
        d.mkdir(parents=True)
        (d / "power1_average").write_text(str(uw))
        (d / "power1_cap").write_text(str(cap))
        (d / "freq1_input").write_text(str(hz))
    ps = bench.PowerSampler(period=0.01, sysfs_root=str(tmp_path)).start()
    time.sleep(0.08)
    per = ps.stop()
    assert len(per) == 2 and per[0]["mean_w"] == 250.0 and per[1]["mean_w"] == 1395.0 and per[1]["n"] >= 3
    out = ps.summary(per, seconds_per_step=0.0422)
    assert out["mean_w"] == 1395.0 and out["cap_w"] == 1400.0 and out["frac_of_cap"] == 0.9964 and out["mean_sclk_mhz"] == 2135
    assert out["joules_per_step"] == round(1395.0 * 0.0422, 2) and "index 1" in out["source"]
    empty = bench.PowerSampler(sysfs_root=str(tmp_path / "nothing")).start()
    assert "error" in empty.summary(empty.stop())
    share = bench.host_cpu_share()
    assert share["os_cpu_count"] >= share["usable"] >= 1
