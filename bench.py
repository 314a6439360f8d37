#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the ViT-B/16 16x224x224 fine-tuning step (forward + backward + gradient
all-reduce + AdamW) on N MI355X GPUs of one node -- BASELINE.json configs[2] (N=1) / configs[3] (N=8).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N ranks itself, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch (32 clips per GPU, resident in HBM): per-step lr
assignment, forward, CE loss, backward (bucketed RCCL all-reduce overlapped), grad-norm, AdamW step, zero_grad.
Rank 0 prints ONE JSON line.

`python bench.py --gpus N` without a launcher (no RANK / WORLD_SIZE in the environment) re-starts itself as N ranks through
`python -m torch.distributed.run` BEFORE anything touches the GPU (the parent only waits for the launcher and exits with its
code; it never initialises HIP, so no process that holds the GPU is ever replaced).

Every performance figure in the line is measured by THIS run.  Fields beside the contract's:
  roofline       dominant kernel gemm_nt: HIP events around every 13th Linear call INSIDE the timed region.  `traffic` (HBM bytes per
                 launch from rocprofv3 PMC passes) cannot be taken inside a run: it comes from the committed profile of this same
                 command and is reported only while the kernel sources still hash to what that profile was taken on, else null.
  from_profiles  what was read from profiles/ (file, source hash match): traffic, the clock held inside the MFMA loops.
  roofline_all   every kernel class, from 3 extra un-timed steps with events around every launch.
  engine_loop    the same step driven by engine.train_one_epoch, i.e. WITH the reference loop's two host syncs per step
                 (engine_for_finetuning.py:71 loss.item(), :102 torch.cuda.synchronize()).
  fwd_only       BASELINE configs[1], timed after the training region.
  half           the step on IEEE-half operands with GradScaler-style loss scaling (the reference's own autocast arithmetic):
                 throughput, and forward AND backward deviation from the reference's fp64 run (golden G11) beside the bf16 mode's.
  precise        split-operand Linears + f32 attention: throughput and the same deviation figures.
  mae_pretrain   BASELINE configs[4] on one GPU: ViT-L/16 encoder on 392 visible tokens + 12-block decoder, tube mask 0.75.
  torch_route    the reference's operator route as stock torch modules on this GPU (calibration).
  cpu_baseline   the oracle on the host cores, last.
  collective     (N > 1) backend, buckets, exposed all-reduce time, per-bucket time, per-rank step times.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU (MI355X_MICROARCH.md: ~2.5 PF dense); the half MFMA rate is the same
PEAK_HBM_GBS = 8000.0
NOMINAL_CLOCK_MHZ = 2400.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (weak scaling: fixed per-GPU work)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: fixed TOTAL clips per step, split over the ranks "
                                                               "(SURVEY 8d cfg4: 256); overrides --batch")
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--precision", default="fast", choices=["fast", "half"], help="operand format of the TIMED region (default: bf16 = the "
                    "headline; half is reported in the `half` object of the default run anyway)")
    ap.add_argument("--drop-path", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline on ALL host cores (a 256-thread pool on a "
                    "batch of 2 oversubscribes: minutes per batch on the pool's 256-core host; off by default)")
    ap.add_argument("--cpu-baseline-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-host-cores leg of the CPU baseline (it is bounded to ~60 s and on by default)")
    ap.add_argument("--linear-schedule", default="auto", choices=["auto", "per-tile", "persistent"], help="world > 1 only: the Linear GEMMs' grid while an "
                    "RCCL kernel may hold CUs (auto = per-tile; parallel.DataParallel)")
    ap.add_argument("--bucket-dtype", default="f32", choices=["f32", "bf16"], help="world > 1 only: format of the gradient buckets on the wire "
                    "(DataParallel(bucket_dtype=...): bf16 halves the bytes of the exchange; default f32)")
    ap.add_argument("--no-rccl-info", action="store_true", help="world > 1: do not collect RCCL's own description of the communicator (NCCL_DEBUG=INFO "
                    "INIT / GRAPH lines of rank 0 in a temporary file: version, channels, transport per peer) into the `collective` object")
    ap.add_argument("--no-live-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip everything after the timed region except cpu_baseline")
    ap.add_argument("--only-extras", default="", help="comma list out of power,roofline_all,engine_loop,fwd_only,half,precise,mae_pretrain,torch_route,vit_small,vit_large "
                                                      "(default: all)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-class table to stderr")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only: rendezvous, one all-reduce, rank 0 prints the world size "
                                                          "(no GPU work; runs over gloo on a CPU-only host -- tests/test_bench_launch.py)")
    ap.add_argument("--graph", type=int, default=-1, help="1: capture the whole step in a HIP graph and replay it (N=1 only); default eager")
    return ap.parse_args()


def self_launch(args):
    """N > 1 without a launcher: become the launcher.  Runs before `import torch` has initialised anything on the GPU."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ or "OMPI_COMM_WORLD_RANK" in os.environ:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


class PowerSampler:
    """Board power, power cap and shader clock of the GPUs visible in sysfs (amdgpu hwmon: power1_average / power1_input in microwatts,
    power1_cap, freq1_input), read by a thread every `period` seconds while a region runs.  Which card is the one under load is decided
    from the samples (the card whose mean power is highest): the container sees every card of the host in sysfs but runs on one.
    Why it is in the bench line: every MFMA kernel of the step runs AT the board's power cap, so the cap, not the
    matrix pipe's nominal rate, is what bounds the step -- the line carries the evidence with every run."""

    def __init__(self, period=0.05, sysfs_root="/sys"):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], None, None
        self.cards = []
        for d in sorted(glob.glob(os.path.join(sysfs_root, "class/drm/card*/device/hwmon/hwmon*"))):
            p = next((os.path.join(d, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))), None)
            if p:
                self.cards.append((d, p))

    def _read(self):
        out = []
        for d, p in self.cards:
            try:
                w = int(open(p).read()) / 1e6
                f = os.path.join(d, "freq1_input")
                out.append((w, int(open(f).read()) / 1e6 if os.path.exists(f) else None))
            except Exception:  # noqa: BLE001
                out.append((None, None))
        return out

    def caps(self):
        caps = []
        for d, _ in self.cards:
            try:
                caps.append(int(open(os.path.join(d, "power1_cap")).read()) / 1e6)
            except Exception:  # noqa: BLE001
                caps.append(None)
        return caps

    def start(self):
        import threading
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(), self._read()))
                self._stop.wait(self.period)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def stop(self, drop_s=0.0):
        """per card: {mean_w, max_w, mean_sclk_mhz, n} over the samples taken at least `drop_s` after start()"""
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        self._th = None
        if not self.samples or not self.cards:
            return None
        t0 = self.samples[0][0]
        keep = [s for t, s in self.samples if t - t0 >= drop_s] or [s for _, s in self.samples]
        per = []
        for c in range(len(self.cards)):
            ws = [s[c][0] for s in keep if s[c][0] is not None]
            fs = [s[c][1] for s in keep if s[c][1] is not None]
            per.append({"mean_w": sum(ws) / len(ws) if ws else None, "max_w": max(ws) if ws else None,
                        "mean_sclk_mhz": sum(fs) / len(fs) if fs else None, "n": len(ws)})
        return per

    def summary(self, per, seconds_per_step=None):
        """the card under load, as one small object for the bench line"""
        if not per:
            return {"error": "no amdgpu hwmon power file readable under /sys/class/drm"}
        loaded = max(range(len(per)), key=lambda i: per[i]["mean_w"] or 0.0)
        c, caps = per[loaded], self.caps()
        cap = caps[loaded] if loaded < len(caps) else None
        out = {"mean_w": round(c["mean_w"], 1) if c["mean_w"] else None, "max_w": c["max_w"], "cap_w": cap,
               "frac_of_cap": round(c["mean_w"] / cap, 4) if (c["mean_w"] and cap) else None,
               "mean_sclk_mhz": round(c["mean_sclk_mhz"]) if c["mean_sclk_mhz"] else None, "samples": c["n"],
               "source": f"sysfs hwmon of the loaded card ({len(per)} cards visible, index {loaded}), sampled every {self.period:g} s during the timed region"}
        if seconds_per_step and c["mean_w"]:
            out["joules_per_step"] = round(c["mean_w"] * seconds_per_step, 2)
        return out


def rccl_summary(log_path):
    """what RCCL says about the communicator of this run: library version, channel count, and how rank 0 reaches its peers (the NCCL_DEBUG=INFO
    lines "Channel 03/0 : 0[0] -> 1[1] via P2P/IPC" etc.: P2P = peer-to-peer over xGMI, SHM / NET = through host memory / the network)"""
    import re
    out = {}
    try:
        import torch
        v = torch.cuda.nccl.version()
        out["version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        out["version_error"] = repr(e)
    if not log_path:
        out["debug_lines"] = "not collected (--no-rccl-info, a dry run, or NCCL_DEBUG already set by the caller)"
        return out
    try:
        import glob
        txt = "".join(open(f, errors="replace").read() for f in sorted(glob.glob(log_path + "*")))
        via = re.findall(r"Channel (\d+)/\d+ : (\d+)\[[^\]]*\] -> (\d+)\[[^\]]*\] .*?via (\S+)", txt)
        out["channels"] = len({c for c, *_ in via}) or None
        out["transports"] = sorted({t for *_, t in via})
        out["peers_of_rank0"] = sorted({int(dst) for _, src, dst, _ in via if src == "0"})
        keep = [ln.split("NCCL INFO", 1)[-1].strip() for ln in txt.splitlines()
                if re.search(r"version|nChannels|Channel 00|Trees|Ring 00|XGMI|xGMI|PCI|threadThresholds|comm .* rank 0", ln)]
        out["lines"] = keep[:12]
        out["transport_of_rank0"] = ("xGMI peer-to-peer" if any(t.startswith("P2P") for t in out["transports"]) else
                                     "host memory / network (no P2P line)" if out["transports"] else "unknown (no Channel lines in the log)")
    except Exception as e:  # noqa: BLE001
        out["debug_error"] = repr(e)
    return out


def flops_per_clip(N, D, L, n_cls=2, k_patch=1536):
    """algorithmic FLOPs (2/MAC, full N^2 attention, no recompute, patch-embed bwd = dW only)."""
    f_patch = 2 * N * k_patch * D
    f_blk = 24 * N * D * D + 4 * N * N * D
    return f_patch + L * f_blk + 2 * D * n_cls, 3 * L * f_blk + 2 * f_patch + 6 * D * n_cls


def csrc_sha16():
    """hash of the kernel sources + C header: ties a figure read from profiles/ to the code it was measured on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "simple_tad_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "tad_mi355x.h")]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def read_profiles():
    """Figures that need a profiler or an ablation build (HBM traffic per gemm_nt launch from rocprofv3 PMC passes; the clock held inside
    the MFMA loops from the in-kernel stamps): read from the latest committed profile, each with the source hash it was taken on."""
    import glob
    out = {"csrc_sha16": csrc_sha16()}
    for key, pat, field in (("gemm_nt_traffic_bytes_per_launch", "r*_summary.json", "gemm_nt_traffic_bytes_per_launch"),
                            ("held_clock_mhz", "r*_clock.json", "held_clock_mhz")):
        try:
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]
            d = json.load(open(latest))
            out[key] = {"value": d.get(field), "file": os.path.relpath(latest, ROOT), "measured_on_csrc_sha16": d.get("csrc_sha16"),
                        "current": d.get("csrc_sha16") == out["csrc_sha16"]}
        except Exception as e:  # noqa: BLE001
            out[key] = {"value": None, "error": repr(e)}
    return out


def host_cpu_share():
    """what the process may actually use of the host: os.cpu_count() counts every core of the machine, the scheduler affinity and the
    cgroup CPU quota say how many of them this job gets (a 1-GPU box of an 8-GPU host is given a share)"""
    n = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        aff = n
    quota = None
    for path, v2 in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            if v2:
                q, per = open(path).read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            else:
                q = float(open(path).read())
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:  # noqa: BLE001
            continue
    mem = None
    try:
        import psutil
        mem = psutil.virtual_memory().available
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                lim = open(path).read().strip()
                if lim != "max":
                    mem = min(mem, int(lim))
                break
            except Exception:  # noqa: BLE001
                continue
    except Exception:  # noqa: BLE001
        pass
    return {"os_cpu_count": n, "affinity": aff, "cgroup_quota_cpus": quota, "usable": int(min(aff, quota) if quota else aff), "mem_available_bytes": mem}


def cpu_baseline(state_dict, frames, reps=3, all_cores=True):
    """Oracle (pure-torch CPU restatement of the reference path) fwd+bwd, B=2, fp32: 1 warm-up + `reps` timed passes on at most 32
    host cores (a larger pool on this small batch oversubscribes) and -- BASELINE.md section 3 -- a second leg on os.cpu_count() threads
    at a batch that pool can use (8 clips), each leg bounded so that the two together stay within about two minutes."""
    import torch
    from oracle import vit_oracle as O
    P = {k: v.detach().float().cpu().requires_grad_() for k, v in state_dict.items()}
    torch.manual_seed(0)
    x = torch.randn(2, 3, frames, 224, 224)
    y = torch.randint(0, 2, (2,))

    def one():
        for p in P.values():
            p.grad = None
        logits = O.forward(x, P, depth=12, num_heads=12, tubelet=2, patch=16)
        torch.nn.functional.cross_entropy(logits, y).backward()

    def timed(threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        one()  # warm-up
        warm = time.perf_counter() - t0
        if warm > 15.0:  # keep the default bench run bounded: a slow host reports the warm-up pass itself
            return 0, warm
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        return reps, (time.perf_counter() - t0) / reps

    share = host_cpu_share()
    ncpu = share["os_cpu_count"]
    n32 = min(ncpu, 32)
    r, dt = timed(n32)
    out = {"value": round(2 / dt, 4), "unit": "clips/sec", "cores": n32, "kind": "port", "host_cores": ncpu, "reps": r, "host_cpu_share": share,
           "sample": f"ViT-B/16 16x224x224 fwd+bwd (CE loss), batch 2, fp32, {r} reps after 1 warm-up, oracle/vit_oracle.py "
                     f"on {n32} of {ncpu} host cores; {dt:.2f} s per batch"}
    if ncpu > n32 and all_cores:
        # BASELINE.md section 3 asks for os.cpu_count() threads.  At batch 2 a 256-thread pool oversubscribes so badly that ONE pass takes
        # minutes (196 s measured in round 4), so this leg runs 8 clips (4 / 2 when the host's free memory is short: ~3.6 GB per clip) in a
        # child process (CPU only: it never touches the GPU) with a passive OpenMP wait policy (spinning workers burn the job's CPU quota),
        # prints every pass as it finishes and is stopped after 100 s: the line carries the last pass that finished.
        import subprocess
        torch.set_num_threads(n32)
        mem = share.get("mem_available_bytes")
        cb = 8 if (mem is None or mem > 64e9) else 4 if mem > 36e9 else 2
        env = {**os.environ, "HIP_VISIBLE_DEVICES": "", "TAD_CPU_CHILD": "1", "OMP_WAIT_POLICY": "PASSIVE", "GOMP_SPINCOUNT": "0", "KMP_BLOCKTIME": "0"}
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(ncpu), "--frames", str(frames), "--batch", str(cb)]
        txt, timed_out = "", False
        try:
            txt = subprocess.run(cmd, capture_output=True, text=True, timeout=100, env=env).stdout
        except subprocess.TimeoutExpired as e:
            timed_out = True
            txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        except Exception as e:  # noqa: BLE001
            out["all_cores"] = {"value": None, "cores": ncpu, "error": repr(e)}
        if "all_cores" not in out:
            passes = []
            for ln in txt.strip().splitlines():
                try:
                    passes.append(json.loads(ln))
                except Exception:  # noqa: BLE001
                    pass
            if passes:
                d = passes[-1]
                out["all_cores"] = {"value": round(cb / d["s_per_batch"], 4), "unit": "clips/sec", "cores": ncpu, "batch": cb, "s_per_batch": round(d["s_per_batch"], 2),
                                    "pass": d["pass"], "passes_finished": len(passes), "stopped_at_100s": timed_out,
                                    "note": f"the same model, {cb} clips per pass on os.cpu_count() = {ncpu} threads (passive OpenMP waits), in a child process; "
                                            f"`value` above is the {n32}-thread pool at batch 2.  This job's CPU share of the host: {share['usable']} "
                                            "(host_cpu_share)"}
            else:
                out["all_cores"] = {"value": None, "cores": ncpu, "batch": cb, "stopped_at_100s": timed_out,
                                    "note": f"not one pass of {cb} clips finished within 100 s on a {ncpu}-thread pool; this job's CPU share of the host is "
                                            f"{share['usable']} cores (host_cpu_share): `value` above is the {n32}-thread pool"}
    return out


def cpu_baseline_child(threads, frames, batch=2):
    """the all-host-cores leg of cpu_baseline, run as `python bench.py --cpu-baseline-child N --batch B` by the parent (which stops it after
    100 s): the same seeded ViT-B/16 (the module surface constructs on the CPU; the oracle does the arithmetic); every pass prints its time"""
    import torch
    import simple_tad_amd as T
    from oracle import vit_oracle as O
    torch.manual_seed(0)
    m = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=frames, tubelet_size=2, final_reduction="fc_norm",
                       drop_path_rate=0.0, init_scale=0.001, use_flash_attn=True)
    P = {k: v.detach().float().requires_grad_() for k, v in m.state_dict().items()}
    torch.manual_seed(0)
    x = torch.randn(batch, 3, frames, 224, 224)
    y = torch.randint(0, 2, (batch,))
    torch.set_num_threads(threads)

    def one():
        for p_ in P.values():
            p_.grad = None
        torch.nn.functional.cross_entropy(O.forward(x, P, depth=12, num_heads=12, tubelet=2, patch=16), y).backward()

    for i in range(2):  # pass 0 is the warm-up, pass 1 the timed one (~36 s each on the GPU box's 16-core share); the parent reports the LAST pass that finished inside its limit
        t0 = time.perf_counter()
        one()
        print(json.dumps({"s_per_batch": time.perf_counter() - t0, "pass": "warm-up" if i == 0 else f"timed {i}", "reps": 1, "threads": threads, "batch": batch}), flush=True)


# --------------------------------------------------------------------------------------------------------------- parity vs golden G11
def parity_vs_golden(T, dev, modes, loss_scale=4096.0):
    """Forward AND backward deviation of each precision mode from the reference's own fp64 run of ViT-B/16 16x224x224, B = 2 (fixture
    data only: the reference never travels to the GPU box).  Gradients, over all 152 tensors: `rms` = error of the stored 256-element
    slice relative to the tensor's RMS, `sqsum` = error of the tensor's sum of squares, `slice` = relative error of the slice itself
    (the first row of a qkv weight is a q row, whose gradient at seeded init is ~100x below the tensor's RMS and ill-conditioned)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_recipe as R
    g = np.load(os.path.join(ROOT, "tests", "golden", "g11_vitb_grads.npz"), allow_pickle=False)

    def rel(a, b):
        a, b = a.double().cpu(), torch.from_numpy(np.asarray(b)).double()
        return ((a - b).norm() / b.norm()).item()

    res = {}
    for mode in modes:
        torch.manual_seed(0)
        m = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                           use_flash_attn=False, init_scale=1.0, drop_path_rate=0.0)
        R.rerandomize_1d(m)
        torch.manual_seed(1)
        xs = torch.randn(2, 3, 16, 224, 224).to(dev)
        m = m.to(dev).train()
        T.set_precision(mode)
        try:
            scale = loss_scale if mode == "half" else 1.0
            f = m.forward_features(xs)
            lg = m.head(f)
            loss = F.cross_entropy(lg, torch.tensor([0, 1], device=dev))
            (loss * scale).backward()
            torch.cuda.synchronize()
        finally:
            T.set_precision("fast")
        e_rms, e_sq, e_sl = [], [], []
        for k, p in m.named_parameters():
            gr = (p.grad / scale).double().cpu()
            head = torch.from_numpy(g["grad." + k + ".head"]).double()
            sq = float(g["grad." + k + ".sqsum"])
            d = (gr.flatten()[: head.numel()] - head).norm()
            e_sl.append((d / head.norm().clamp_min(1e-300)).item())
            e_rms.append((d / (head.numel() ** 0.5 * max((sq / gr.numel()) ** 0.5, 1e-300))).item())
            e_sq.append(abs((gr ** 2).sum().item() - sq) / max(sq, 1e-300))
        r3 = lambda v: float(f"{v:.3e}")  # noqa: E731
        res[mode] = {"features_rel_l2": r3(rel(f.detach(), g["features"])), "logits_rel_l2": r3(rel(lg.detach(), g["logits"])),
                     "loss_abs": r3(abs(loss.item() - float(g["loss"]))),
                     "grad_rms": {"median": r3(float(np.median(e_rms))), "worst": r3(max(e_rms))},
                     "grad_sqsum": {"median": r3(float(np.median(e_sq))), "worst": r3(max(e_sq))},
                     "grad_slice": {"median": r3(float(np.median(e_sl))), "worst": r3(max(e_sl)), "tensors_over_1e-3": int(sum(v > 1e-3 for v in e_sl)),
                                    "tensors": len(e_sl)}}
        del m
        torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child(args.cpu_baseline_child, args.frames, batch=args.batch if args.batch != 32 else 2)
        return
    self_launch(args)

    import torch
    import simple_tad_amd as T
    from simple_tad_amd import engine as E
    from simple_tad_amd import kernels as K
    from simple_tad_amd import _lib
    from simple_tad_amd.parallel import DataParallel, init_distributed_mode

    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.no_rccl_info and not args.dry_run and "NCCL_DEBUG" not in os.environ:
        # RCCL describes the communicator it builds (version, channels, the transport to every peer: P2P over xGMI, or through host memory /
        # PCIe) when NCCL_DEBUG=INFO is set BEFORE the process group exists: rank 0's lines go to a file that the `collective` object quotes
        import tempfile
        rccl_log = os.path.join(tempfile.gettempdir(), f"tad_rccl_{os.getpid()}.log")
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,ENV", NCCL_DEBUG_FILE=rccl_log)
    distributed, rank, world, local = init_distributed_mode()
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        t = torch.ones(1, device="cuda" if (torch.cuda.is_available() and torch.distributed.is_initialized()
                                             and torch.distributed.get_backend() == "nccl") else "cpu")
        if distributed:
            torch.distributed.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "allreduce_of_ones": t.item(),
                              "backend": torch.distributed.get_backend() if distributed else None,
                              "flags": {"linear_schedule": args.linear_schedule, "bucket_dtype": args.bucket_dtype, "global_batch": args.global_batch,
                                        "per_gpu_batch": (args.global_batch // world) if args.global_batch else args.batch,
                                        "scaling": "strong" if args.global_batch else "weak"},
                              "launched_by": "self" if os.environ.get("TORCHELASTIC_RUN_ID") else "direct"}), flush=True)
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the MI355X path has no CPU fallback)"
    if os.environ.get("TAD_DIST_BACKEND") == "gloo":  # debugging aid: several ranks share the visible GPUs
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    info = K.device_info()
    T.set_precision(args.precision)

    torch.manual_seed(0)  # identical init on every rank
    model = T.create_model(args.model, pretrained=False, num_classes=2, all_frames=args.frames, tubelet_size=2,
                           final_reduction="fc_norm", drop_path_rate=args.drop_path if args.mode == "train" else 0.0,
                           init_scale=0.001, use_flash_attn=True).to(dev)
    sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()} if rank == 0 else None
    D, L = model.embed_dim, model.get_num_layers()
    ntok = model.patch_embed.num_patches
    f_fwd, f_fb = flops_per_clip(ntok, D, L)
    n_params = sum(p.numel() for p in model.parameters())

    torch.manual_seed(0 + rank)  # per-rank data seed (run_class_finetuning.py:222)
    B = args.batch
    if args.global_batch:
        assert args.global_batch % world == 0, "--global-batch must be a multiple of the number of ranks"
        B = args.global_batch // world
    x = torch.randn(B, 3, args.frames, 224, 224, device=dev)
    y = torch.randint(0, 2, (B,), device=dev)
    total_steps = args.steps + args.warmup + 512  # (+ the un-timed extra steps after the region: power, roofline_all, engine_loop, half, precise)
    lr_sched = E.cosine_scheduler(5e-4 * B * world / 256, 1e-6, 1, max(total_steps, 2), warmup_epochs=0)

    dp = opt = scaler = None
    if args.mode == "train":
        model.train()
        dp = DataParallel(model, bucket_mb=64.0, linear_schedule=None if args.linear_schedule == "auto" else args.linear_schedule,
                          bucket_dtype=args.bucket_dtype)
        if distributed:
            dp.enable_timing()
        opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
        scaler = E.NativeScalerWithGradNormCount(dp)
        crit = torch.nn.CrossEntropyLoss()
        params = [p for p in model.parameters()]
        dp.zero_grad()

        def step(it):
            for g in opt.param_groups:
                g["lr"] = lr_sched[it] * g["lr_scale"]
            loss = crit(dp(x), y)
            scaler(loss, opt, parameters=params, update_grad=True)
            dp.zero_grad()
            return loss

        use_graph = (args.graph == 1)  # measured: the step is not launch-bound (eager 61.8 ms vs graph 62.3 ms), so eager is the default
        if use_graph and world == 1:
            # Launch-bound stretches (hundreds of short launches per step) are removed by capturing the whole step -- forward,
            # loss, backward, grad-norm, AdamW, zero_grad -- into ONE HIP graph and replaying it.  The learning rate lives in a
            # device tensor per param group (capturable AdamW) that is refreshed before each replay, so the schedule still applies.
            # (torch's fused AdamW does not bump Parameter._version: the scaler's invalidate_weight_cache() call after its step drops
            # FusedAdamW's operand mirrors, so the captured forward re-casts the weights inside the graph.)
            opt = torch.optim.AdamW([{"params": g["params"], "weight_decay": g["weight_decay"], "lr_scale": g["lr_scale"],
                                      "lr": torch.tensor(float(g["lr"]), device=dev)} for g in opt.param_groups],
                                    betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=True, fused=True)
            from simple_tad_amd import ops as _ops
            _ops.invalidate_weight_cache()
            lr_table = torch.tensor(lr_sched, dtype=torch.float32, device=dev)
            scales = [g["lr_scale"] for g in opt.param_groups]

            def eager_body():
                loss = crit(dp(x), y)
                scaler(loss, opt, parameters=params, update_grad=True)
                dp.zero_grad()
                return loss

            # warm-up AND capture on one stream: kernels.workspace is per (device, stream), so the scratch buffers the graph bakes in are
            # allocated by the warm-up, outside the graph's private pool; they are held for as long as the graph lives
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_body()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_loss = eager_body()
            graph_scratch = K.workspace_refs(dev, side)  # noqa: F841  (kept alive with the graph)

            def step(it):  # noqa: F811
                for g, sc in zip(opt.param_groups, scales):
                    g["lr"].copy_(lr_table[it] * sc)
                graph.replay()
                return static_loss
    else:
        model.eval()

        def step(it):
            with torch.no_grad():
                return model(x)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step(it)
    if dp is not None and dp._timing is not None:
        dp._timing["steps"].clear()  # (the warm-up steps' records)
    prof = None
    if rank == 0 and not args.no_live_profile:
        prof = K.LaunchProfiler(only=["gemm_nt"], stride=13)
        K.set_profiler(prof)
    power = PowerSampler().start() if rank == 0 else None  # (a thread that reads two sysfs files every 50 ms: no GPU work, no sync)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        last = step(args.warmup + it)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    power_cards = power.stop() if power is not None else None
    K.set_profiler(None)
    collective = None
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
        own = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        torch.distributed.all_gather(own, torch.tensor([dt_own], dtype=torch.float64, device=dev))
        per_rank = [1e3 * float(o.item()) / args.steps for o in own]
        collective = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                      "allreduce_bytes_per_step": dp.exchange_bytes_per_step() if dp is not None else 0,
                      "bucket_dtype": dp.bucket_dtype if dp is not None else None,
                      "buckets": len(dp.buckets) if dp is not None else 0,
                      "rank_ms_per_step": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3), "all": [round(v, 3) for v in per_rank]},
                      "linear_schedule": (dp.linear_schedule if dp is not None else None),
                      "linear_schedule_note": "per-tile = one workgroup per tile (DataParallel's default at world > 1), persistent = the single-GPU "
                                              "schedule; bench.py --linear-schedule {per-tile,persistent} A/Bs the two"}
        ts = dp.timing_summary() if dp is not None else None
        if ts is not None:
            collective.update(ts)
        collective["rccl"] = rccl_summary(rccl_log)
    loss_val = float(last.float().mean().item()) if args.mode == "train" else float("nan")

    if rank != 0:
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    clips_per_s = B * world * args.steps / dt
    fl = f_fb if args.mode == "train" else f_fwd
    out = {
        "metric": "clips/sec (16x224^2 ViT-B) " + ("fwd+bwd" if args.mode == "train" else "forward"),
        "value": round(clips_per_s, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None,
        "dtype": "bf16" if args.precision == "fast" else "f16", "data": "synthetic",
        "config": {"workload": f"{args.model} {args.frames}x224x224, {B} clips/GPU, " +
                               ("fwd+bwd+AdamW fine-tune step with CE loss on synthetic labels (BASELINE configs[2]/[3])"
                                if args.mode == "train" else "forward only (BASELINE configs[1])"),
                   "global_batch": B * world, "per_gpu_batch": B, "tokens_per_clip": ntok, "parallelism": f"dp{world}",
                   "drop_path": args.drop_path if args.mode == "train" else 0.0, "residual_stream": "f32",
                   "operands": "bf16" if args.precision == "fast" else "f16 + loss scaling",
                   "algorithmic_gflop_per_clip": round(fl / 1e9, 2), "parameters": n_params},
        "frac_of_bf16_mfma_roofline": round(clips_per_s * fl / world / (PEAK_BF16_TFLOPS * 1e12), 4),
        "loss": loss_val,
        "device": info,
    }
    if collective is not None:
        out["collective"] = collective
    try:
        out["power"] = power.summary(power_cards, seconds_per_step=dt / args.steps)
        if out["power"].get("joules_per_step"):
            out["power"]["picojoules_per_algorithmic_flop"] = round(out["power"]["joules_per_step"] / (B * fl) * 1e12, 3)
    except Exception as e:  # noqa: BLE001
        out["power"] = {"error": repr(e)}
    fp = read_profiles()
    out["from_profiles"] = fp
    held = fp.get("held_clock_mhz", {})
    if held.get("value") and held.get("current"):
        out["from_profiles"]["frac_of_roofline_at_held_clock"] = round(out["frac_of_bf16_mfma_roofline"] * NOMINAL_CLOCK_MHZ / float(held["value"]), 4)
    if prof is not None:
        summ = prof.summary()
        g = summ.get("gemm_nt")
        if g and g["ms"] > 0:
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            tr = fp.get("gemm_nt_traffic_bytes_per_launch", {})
            out["roofline"] = {"kernel": "gemm_nt_kernel (16-bit-operand MFMA GEMM, all Linear fwd / input-grad launches)", "bound": "mfma",
                               "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               "traffic": round(tr["value"]) if (tr.get("value") and tr.get("current")) else None,
                               "traffic_source": (tr.get("file") if tr.get("current") else
                                                  "none: the committed PMC profile was taken on other kernel sources (from_profiles)"),
                               # a Linear call is one kernel launch, or two under a split plan: times and flops are per KERNEL
                               # launch, as rocprofv3 counts them
                               "calls": prof.seen.get("gemm_nt", g["calls"]), "sampled_calls": g["calls"], "sampled_launches": g["launches"],
                               "launches": round(prof.seen.get("gemm_nt", g["calls"]) * g["launches"] / max(g["calls"], 1)),
                               "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
                               "gflop_per_launch": round(g["flops"] / g["launches"] / 1e9, 2),
                               # both operands + the output once, + the f32 residual / 16-bit pre-activation the epilogue reads or writes
                               # (kernels.linear_fwd / linear_bwd_input), averaged over the sampled launches like `achieved`
                               "algorithmic_bytes": round(g["bytes"] / g["launches"])}
            if out["roofline"]["traffic"]:
                out["roofline"]["traffic_over_algorithmic"] = round(out["roofline"]["traffic"] / max(out["roofline"]["algorithmic_bytes"], 1), 3)

    want = set(filter(None, args.only_extras.split(","))) or {"roofline_all", "engine_loop", "fwd_only", "half", "precise", "mae_pretrain", "torch_route",
                                                                "vit_small", "vit_large", "inference_b1", "power"}
    extras = world == 1 and not args.no_extras and args.mode == "train" and args.graph != 1
    if args.precision != "fast":  # the other modes' objects compare against the bf16 headline: only the per-class table makes sense here
        want &= {"roofline_all"}
    it_next = args.warmup + args.steps

    def timed_steps(n_warm, n):
        nonlocal it_next
        for _ in range(n_warm):
            step(it_next)
            it_next += 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            step(it_next)
            it_next += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n

    def class_table(summ, nsteps, batch):
        ra = {}
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            if v["ms"] <= 0:
                continue
            ent = {"launches_per_step": round(v["launches"] / nsteps, 1), "ms_per_step": round(v["ms"] / nsteps, 3)}
            if v["flops"] > 0:
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
                ent.update({"bound": "mfma", "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)})
            else:
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                ent.update({"bound": "hbm", "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
            ra[k] = ent
        return ra

    if extras and "power" in want:
        # ---- board power in steady state: the hwmon reading is a moving average over about a second, so the 0.9 s timed region above reads low
        # (`power.timed_region`); here the same step runs back to back for ~3 s and the first 1.2 s of samples are dropped
        try:
            ps = PowerSampler(period=0.02).start()
            n_p = max(60, int(3.0 / max(dt / args.steps, 1e-3)))
            dtp = timed_steps(0, n_p)
            pw = ps.summary(ps.stop(drop_s=1.2), seconds_per_step=dtp)
            pw["steps"], pw["ms_per_step"] = n_p, round(1e3 * dtp, 3)
            if pw.get("joules_per_step"):
                pw["picojoules_per_algorithmic_flop"] = round(pw["joules_per_step"] / (B * fl) * 1e12, 3)
            pw["timed_region"] = out.get("power")
            pw["reading"] = ("the whole training step draws this share of the board's power cap: the cap, not the matrix pipe's nominal rate, bounds every MFMA "
                             "kernel of the step (DESIGN.md section 3a; per kernel class: tools/power_probe.py -> profiles/r06_power.txt)")
            out["power"] = pw
            if isinstance(out.get("roofline"), dict) and pw.get("frac_of_cap"):
                # the nominal-peak fraction above is measured on a board that runs this kernel (and the whole step) at its power cap: DESIGN.md section 3a
                out["roofline"]["board_power_frac_of_cap_during_step"] = pw["frac_of_cap"]
        except Exception as e:  # noqa: BLE001
            out["power"] = {"error": repr(e), "timed_region": out.get("power")}

    if extras and "roofline_all" in want:
        # ---- every kernel class, HIP events around every launch of 3 extra steps (the events add ~20 us of queue time per launch, so
        # these steps are NOT part of `value`)
        try:
            pall = K.LaunchProfiler(only=None, stride=1)
            K.set_profiler(pall)
            nrf = 3
            for i in range(nrf):
                step(it_next)
                it_next += 1
            K.set_profiler(None)
            ra = class_table(pall.summary(), nrf, B)
            out["roofline_all"] = ra
            if "adamw" in ra:  # the step with the AdamW update included (`value`) and excluded
                ms_wo = out["ms_per_step"] - ra["adamw"]["ms_per_step"]
                out["excluding_adamw"] = {"ms_per_step": round(ms_wo, 3), "value": round(B * 1e3 / ms_wo, 2), "unit": "clips/sec",
                                          "note": "ms_per_step minus the fused AdamW launch (one HBM-bound kernel per step, roofline_all.adamw)"}
            if args.breakdown:
                for k, e in ra.items():
                    print(f"   {k:16s} {e['launches_per_step']:7.1f} launches/step {e['ms_per_step']:9.3f} ms/step  {e['achieved']:9.1f} {e['unit']}", file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            K.set_profiler(None)
            out["roofline_all"] = {"error": repr(e)}

    if extras and "engine_loop" in want:
        # ---- the reference's loop shape: engine.train_one_epoch = per-step schedule assignment, loss.item() (host sync), backward,
        # optimizer, zero_grad, torch.cuda.synchronize() (host sync), meters -- row (a)12 of the hot-path table
        try:
            nel = 2 + args.steps
            loader = [(x, y)] * nel
            t1 = [None]

            def log(epoch, i, stats):
                if i == 1:  # two warm-up iterations
                    torch.cuda.synchronize()
                    t1[0] = time.perf_counter()
            E.train_one_epoch(dp, crit, loader, opt, dev, 0, scaler, max_norm=0, start_steps=it_next,
                              lr_schedule_values=[v for v in lr_sched], num_training_steps_per_epoch=nel, log=log)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t1[0]) / (nel - 2)
            it_next += nel
            out["engine_loop"] = {"value": round(B / dte, 2), "unit": "clips/sec", "ms_per_step": round(1e3 * dte, 3), "steps": nel - 2,
                                  "host_syncs_per_step": 2, "through": "simple_tad_amd.engine.train_one_epoch (engine_for_finetuning.py:24-140)"}
        except Exception as e:  # noqa: BLE001
            out["engine_loop"] = {"error": repr(e)}

    if extras and "fwd_only" in want:
        # ---- BASELINE configs[1] (eval, no_grad, same batch), timed after the training region
        try:
            model.eval()
            with torch.no_grad():
                for _ in range(5):
                    model(x)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nf = 20
                for _ in range(nf):
                    model(x)
                torch.cuda.synchronize()
                dtf = (time.perf_counter() - t1) / nf
            out["fwd_only"] = {"value": round(B / dtf, 2), "unit": "clips/sec", "ms_per_step": round(1e3 * dtf, 3), "steps": nf, "warmup": 5,
                               "frac_of_bf16_mfma_roofline": round(B / dtf * f_fwd / (PEAK_BF16_TFLOPS * 1e12), 4),
                               "workload": "BASELINE configs[1]: forward only, eval, no_grad, 32 clips",
                               "algorithmic_gflop_per_clip": round(f_fwd / 1e9, 2)}
            model.train()
        except Exception as e:  # noqa: BLE001
            out["fwd_only"] = {"error": repr(e)}

    # ---- the other precision modes: throughput of the SAME training step, and forward + backward deviation of every mode from the
    # reference's fp64 run (tests/golden/g11: the real reference model on ViT-B/16 16x224x224, B = 2)
    par = None
    if extras and ("half" in want or "precise" in want):
        try:
            par = parity_vs_golden(T, dev, ["fast"] + [m for m in ("half", "precise") if m in want])
        except Exception as e:  # noqa: BLE001
            par = {"error": repr(e)}
    against = "tests/golden/g11_vitb_grads.npz: reference modeling_finetune.vit_base_patch16_224, fp64 fwd + CE loss + bwd, B=2, seed-0 init"
    if extras and "half" in want:
        try:
            # (the fused optimizer's operand mirrors are bf16: in half mode they are not served and the forward re-casts the weights,
            #  which a half-mode optimizer built from the start avoids -- tools/exp_precision.py measures that configuration)
            from simple_tad_amd.optim import FusedAdamW
            T.set_precision("half")
            try:
                opt_h = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
                assert isinstance(opt_h, FusedAdamW)
                sc_h = E.NativeScalerWithGradNormCount(dp)
                opt_keep, sc_keep = opt, scaler
                #  the half-mode optimizer steps on the SAME flat parameter buffer with fresh moments: the headline
                # optimizer's parameters are put back afterwards, so its moments, step count and mirrors stay one consistent state
                param_keep = opt.flat_param.clone()
                opt, scaler = opt_h, sc_h
                ra_h = None
                try:
                    dth = timed_steps(3, 10)
                    try:  # the same three un-timed event steps as the bf16 table: where the half step's time goes, class by class
                        pall_h = K.LaunchProfiler(only=None, stride=1)
                        K.set_profiler(pall_h)
                        for _ in range(3):
                            step(it_next)
                            it_next += 1
                        K.set_profiler(None)
                        ra_h = class_table(pall_h.summary(), 3, B)
                    except Exception as e:  # noqa: BLE001
                        K.set_profiler(None)
                        ra_h = {"error": repr(e)}
                finally:
                    opt, scaler = opt_keep, sc_keep
                    opt.flat_param.copy_(param_keep)
                    del param_keep
                hs = {"value": round(B / dth, 2), "unit": "clips/sec", "ms_per_step": round(1e3 * dth, 3), "steps": 10, "warmup": 3,
                      "frac_of_f16_mfma_roofline": round(B / dth * f_fb / (PEAK_BF16_TFLOPS * 1e12), 4), "operands": "IEEE half, f32 accumulation",
                      "loss_scale": sc_h.state_dict()["scale"], "skipped_steps": sc_h.skipped_steps,
                      "vs_value": round((B / dth) / out["value"], 4), "roofline_all": ra_h,
                      "why_slower": "the chip holds a lower clock inside the IEEE-half MFMA loops (profiles/r05_clock_f16.json against "
                                    "r05_clock.json: -6...-7 % in the GEMM K loops, -4 % in attention): DESIGN.md section 4"}
            finally:
                T.set_precision("fast")
                from simple_tad_amd import ops as _ops
                _ops.invalidate_weight_cache()
                opt.refresh_mirrors()  # (the half-mode optimizer moved the parameters: the bf16 operand copies are re-derived)
            out["half"] = hs
        except Exception as e:  # noqa: BLE001
            T.set_precision("fast")
            out["half"] = {"error": repr(e)}
        if isinstance(par, dict) and "half" in par:
            out["half"].update({"deviation": par["half"], "bf16_mode_deviation": par.get("fast"), "tolerance": 1e-3, "against": against})
    if extras and "precise" in want:
        try:
            from simple_tad_amd.modeling_finetune import DropPath
            saved = [(b.drop_path, b.drop_path.drop_prob) for b in model.blocks if isinstance(b.drop_path, DropPath)]
            for dpm, _ in saved:
                dpm.drop_prob = 0.0   # (the precise mode has no drop-path; it does not change the work per step)
            T.set_precision("precise")
            try:
                dtp = timed_steps(2, 3)
            finally:
                T.set_precision("fast")
                for dpm, pr in saved:
                    dpm.drop_prob = pr
            out["precise"] = {"clips_per_s": round(B / dtp, 2), "ms_per_step": round(1e3 * dtp, 3), "steps": 3, "warmup": 2,
                              "frac_of_bf16_mfma_roofline": round(B / dtp * f_fb / (PEAK_BF16_TFLOPS * 1e12), 4)}
        except Exception as e:  # noqa: BLE001
            T.set_precision("fast")
            out["precise"] = {"error": repr(e)}
        if isinstance(par, dict) and "precise" in par:
            out["precise"].update({"deviation": par["precise"], "logits_rel_l2": par["precise"]["logits_rel_l2"],
                                   "features_rel_l2": par["precise"]["features_rel_l2"], "fast_mode_deviation": par.get("fast"),
                                   "tolerance": 1e-3, "against": against,
                                   "reference_bf16_autocast_deviation": {"features_rel_l2": 3.6e-3, "logits_rel_l2": 4.2e-3,
                                                                         "source": "BASELINE.md section 4"}})
    if isinstance(par, dict) and "error" in par:
        out["parity_error"] = par["error"]
    # ---- north_star's bar (1) is "within 1e-3 of the reference": `value` is the bf16 mode north_star names for the MFMA tiles, and its
    # measured deviation is stated in `config`; the mode that IS inside 1e-3 (IEEE-half operands + loss scaling, the reference's own
    # autocast + GradScaler recipe) carries its throughput here, at the top level of the line
    if isinstance(par, dict) and isinstance(par.get("fast"), dict):
        pf = par["fast"]
        out["config"]["headline_mode_deviation"] = {"mode": "fast (bf16 operands)", "logits_rel_l2": pf["logits_rel_l2"], "features_rel_l2": pf["features_rel_l2"],
                                                    "grad_rms_worst": pf["grad_rms"]["worst"], "tolerance": 1e-3, "within_tolerance": False,
                                                    "see": "value_within_tolerance"}
    hv = out.get("half")
    if isinstance(hv, dict) and "value" in hv and isinstance(hv.get("deviation"), dict):
        dv = hv["deviation"]
        out["value_within_tolerance"] = {"mode": "half", "value": hv["value"], "unit": "clips/sec", "ms_per_step": hv["ms_per_step"], "vs_value": hv["vs_value"],
                                         "logits_rel_l2": dv["logits_rel_l2"], "features_rel_l2": dv["features_rel_l2"],
                                         "grad_rms_worst": dv["grad_rms"]["worst"], "tolerance": 1e-3,
                                         "within_tolerance": bool(dv["logits_rel_l2"] <= 1e-3 and dv["features_rel_l2"] <= 1e-3 and dv["grad_rms"]["worst"] <= 1e-3),
                                         "frac_of_bf16_mfma_roofline": hv["frac_of_f16_mfma_roofline"], "against": against}

    if extras and "mae_pretrain" in want:
        # ---- BASELINE configs[4] on one GPU: pretrain_videomae_large_patch16_224 (ViT-L/16 encoder on the 392 visible tokens, 12-block
        # decoder on 1568, jobs/dapt/pretrain_capdata_large.sh:33-36), tube mask 0.75, 32 clips, fwd + bwd + fused AdamW
        try:
            out["mae_pretrain"] = mae_step(T, E, K, dev, class_table, batch=B)
        except Exception as e:  # noqa: BLE001
            K.set_profiler(None)
            out["mae_pretrain"] = {"error": repr(e)}

    # ---- the other two models the reference ships fine-tune jobs for (jobs/finetune/VideoMAE-S_DoTA.sh, VideoMAE-L_D2K.sh; factories
    # modeling_finetune.py:338-388): the same 32-clip fwd + bwd + AdamW step, with their own per-class tables
    for key, name in (("vit_small", "vit_small_patch16_224"), ("vit_large", "vit_large_patch16_224")):
        if extras and key in want and args.model == "vit_base_patch16_224" and args.frames == 16:
            try:
                out[key] = finetune_step(T, E, K, dev, class_table, name, batch=B, drop_path=args.drop_path)
            except Exception as e:  # noqa: BLE001
                K.set_profiler(None)
                out[key] = {"error": repr(e)}

    # ---- the reference's ONLY published performance figure (test_efficiency.py:16-17,174-194): batch-1 forward
    # windows/s of ViT-S / B / L on one 16x224x224 clip under fp16 autocast
    if extras and "inference_b1" in want:
        try:
            out["inference_b1"] = inference_b1(T, K, dev)
        except Exception as e:  # noqa: BLE001
            T.set_precision("fast")
            out["inference_b1"] = {"error": repr(e)}

    # ---- torch_route: the reference's own operator route (stock torch modules under bf16 autocast, torch's fused attention,
    # torch.optim.AdamW; tools/bench_torch_eager.py) for the same workload on THIS GPU, beside `value`.  Calibration only.
    if extras and "torch_route" in want and args.model == "vit_base_patch16_224" and args.frames == 16:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("bench_torch_eager", os.path.join(ROOT, "tools", "bench_torch_eager.py"))
            bte = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bte)
            r = bte.run("sdpa", B, steps=5, warmup=3, device=dev)
            r["speedup_of_value"] = round(out["value"] / r["clips_per_s"], 2)
            out["torch_route"] = r
        except Exception as e:  # noqa: BLE001
            out["torch_route"] = {"error": repr(e)}

    if world == 1 and not args.no_cpu_baseline and args.model == "vit_base_patch16_224":
        try:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, args.frames, all_cores=not args.no_cpu_all_cores)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def inference_b1(T, K, dev, iters=1000):
    """test_efficiency.py's measurement for the three VideoMAE fine-tune factories: ONE clip [1, 3, 16, 224, 224] f32 on the device, eval,
    no_grad, forward only, 1 warm-up + `iters` = 1000 timed calls, in the `half` mode (IEEE-half operands = the reference's fp16 autocast).
    Two figures per model: eager launches and HIP-graph replay.  Unlike the reference's loop (time.time() around each un-synchronised call: it
    times the enqueue while the queue has room), the 1000 calls here are bracketed by torch.cuda.synchronize(), i.e. this is device throughput."""
    import torch
    from simple_tad_amd import ops as _ops
    published = {"vit_small_patch16_224": 95.0, "vit_base_patch16_224": 94.0, "vit_large_patch16_224": 34.0}
    res = {"unit": "windows/sec", "batch": 1, "iters": iters, "precision": "half (IEEE-half operands, f32 accumulation) = the reference's fp16 autocast",
           "timing": "1 warm-up + 1000 forward calls between two torch.cuda.synchronize() (the reference's loop has no device sync)",
           "published_on": "NVIDIA A100 MIG 1/2 GPU (figs/results.png, test_efficiency.py): different hardware, context only", "models": {}}
    T.set_precision("half")
    try:
        for name, pub in published.items():
            torch.manual_seed(0)
            m = T.create_model(name, pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, fc_drop_rate=0.0, drop_rate=0.0, drop_path_rate=0.1,
                               attn_drop_rate=0.0, final_reduction="fc_norm", init_scale=0.001, use_flash_attn=True).to(dev).eval()
            x = torch.randn(1, 3, 16, 224, 224, device=dev)
            ent = {}
            with torch.no_grad():
                for _ in range(3):
                    m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    y = m(x)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                ent["eager"] = {"value": round(1.0 / dt, 1), "ms_per_window": round(1e3 * dt, 3)}
                # HIP-graph replay of the same forward (batch 1 is launch-bound: ~150 short launches per window).  Warm-up and capture on ONE
                # side stream, so the scratch buffers the graph bakes in exist before the capture and are held beside it (kernels.workspace)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        m(x)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    y = m(x)
                keep = (K.workspace_refs(dev, side), _ops.cached_weight_tensors())  # noqa: F841
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    g.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                ent["graph"] = {"value": round(1.0 / dt, 1), "ms_per_window": round(1e3 * dt, 3)}
                assert bool(torch.isfinite(y).all())
            ent["published"] = pub
            ent["vs_published"] = {"eager": round(ent["eager"]["value"] / pub, 2), "graph": round(ent["graph"]["value"] / pub, 2),
                                   "note": "different hardware (1/2 A100 vs one MI355X), context only"}
            res["models"][name] = ent
            del g, keep, m, x, y
            K.release_workspace(dev, side)
            torch.cuda.empty_cache()
    finally:
        T.set_precision("fast")
        _ops.invalidate_weight_cache()
    return res


def finetune_step(T, E, K, dev, class_table, model_name, batch=32, steps=8, warmup=3, drop_path=0.1):
    """the headline's step (forward + CE loss + backward + fused AdamW with layer decay, synthetic clips and labels) for another factory of
    modeling_finetune.py:338-398"""
    import torch
    from simple_tad_amd.parallel import DataParallel
    torch.manual_seed(0)
    model = T.create_model(model_name, pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                           drop_path_rate=drop_path, init_scale=0.001, use_flash_attn=True).to(dev).train()
    dp = DataParallel(model, bucket_mb=64.0)
    opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
    scaler = E.NativeScalerWithGradNormCount(dp)
    crit = torch.nn.CrossEntropyLoss()
    params = list(model.parameters())
    x = torch.randn(batch, 3, 16, 224, 224, device=dev)
    y = torch.randint(0, 2, (batch,), device=dev)
    dp.zero_grad()

    def step():
        loss = crit(dp(x), y)
        scaler(loss, opt, parameters=params)
        dp.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nprof = 2
    pall = K.LaunchProfiler(only=None, stride=1)
    K.set_profiler(pall)
    for _ in range(nprof):
        step()
    K.set_profiler(None)
    ra = class_table(pall.summary(), nprof, batch)
    D, L, ntok = model.embed_dim, model.get_num_layers(), model.patch_embed.num_patches
    _, f_fb = flops_per_clip(ntok, D, L)
    res = {"value": round(batch / dt, 2), "unit": "clips/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "workload": f"{model_name} 16x224x224, {batch} clips, fwd+bwd+AdamW fine-tune step with CE loss on synthetic labels, drop_path {drop_path}",
           "embed_dim": D, "depth": L, "heads": model.num_heads, "algorithmic_gflop_per_clip": round(f_fb / 1e9, 2),
           "frac_of_bf16_mfma_roofline": round(batch * f_fb / dt / (PEAK_BF16_TFLOPS * 1e12), 4), "loss": float(last.detach()),
           "parameters": sum(p.numel() for p in params), "roofline_all": ra}
    del model, dp, opt, scaler, x
    torch.cuda.empty_cache()
    return res


def mae_step(T, E, K, dev, class_table, batch=32, steps=8, warmup=3, model_name="pretrain_videomae_large_patch16_224", decoder_depth=12,
             mask_ratio=0.75):
    """One MAE pre-training step = tube mask -> reconstruction target -> encoder on the visible tokens -> decoder -> MSE -> backward ->
    fused AdamW (engine_for_pretraining.py:51-71 around modeling_pretrain.py:278-291)."""
    import numpy as np
    import torch
    import simple_tad_amd.modeling_pretrain  # noqa: F401  (registers the factories)
    from simple_tad_amd import engine_pretrain as EP, ops
    from simple_tad_amd.masking_generator import TubeMaskingGenerator
    from simple_tad_amd.parallel import DataParallel
    torch.manual_seed(0)
    model = T.create_model(model_name, pretrained=False, drop_path_rate=0.0, decoder_depth=decoder_depth).to(dev).train()
    dp = DataParallel(model)
    opt = E.create_optimizer(dp, lr=3e-4, weight_decay=0.05, betas=(0.9, 0.95))
    scaler = E.NativeScalerWithGradNormCount(dp)
    np.random.seed(0)
    gen = TubeMaskingGenerator((8, 14, 14), mask_ratio)
    x = torch.randn(batch, 3, 16, 224, 224, device=dev)
    nprof = 2
    masks = [torch.from_numpy(np.stack([gen() for _ in range(batch)])).to(dev).bool() for _ in range(steps + warmup + nprof)]
    n_mask = gen.total_masks
    params = list(model.parameters())

    def step(i):
        labels = EP.reconstruction_target(x, masks[i], 16, 2, True, n_mask)
        loss = ops.MseLossFn.apply(dp(x, masks[i], num_masked=n_mask), labels)
        dp.zero_grad()
        scaler(loss, opt, parameters=params)
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pall = K.LaunchProfiler(only=None, stride=1)
    K.set_profiler(pall)
    for i in range(nprof):
        step(warmup + steps + i)
    K.set_profiler(None)
    ra = class_table(pall.summary(), nprof, batch)
    enc, dec = model.encoder, model.decoder
    n_vis = 1568 - n_mask
    blk = lambda n, d: 24.0 * n * d * d + 4.0 * n * n * d  # noqa: E731
    f_fwd = (2.0 * 1568 * 1536 * enc.embed_dim + len(enc.blocks) * blk(n_vis, enc.embed_dim)
             + 2.0 * n_vis * enc.embed_dim * dec.embed_dim + len(dec.blocks) * blk(1568, dec.embed_dim) + 2.0 * n_mask * dec.embed_dim * 1536)
    f_step = 3.0 * f_fwd - 2.0 * 1568 * 1536 * enc.embed_dim  # patch-embed backward is dW only
    res = {"value": round(batch / dt, 2), "unit": "clips/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup,
           "workload": f"{model_name}, decoder depth {decoder_depth}, 16x224x224, {batch} clips, tube mask {mask_ratio}: {n_vis} visible / "
                       f"{n_mask} masked tokens (BASELINE configs[4] on one GPU)",
           "algorithmic_gflop_per_clip": round(f_step / 1e9, 1), "frac_of_bf16_mfma_roofline": round(batch * f_step / dt / (PEAK_BF16_TFLOPS * 1e12), 4),
           "loss": float(last.detach()), "parameters": sum(p.numel() for p in params), "roofline_all": ra}
    del model, dp, opt
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
