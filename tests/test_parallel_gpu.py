"""world_size-2 run of the REAL training step on the GPU: two processes share cuda:0 and exchange gradients over gloo (RCCL needs one
GPU per rank, the box has one).  Checks the N > 1 path end to end on the HIP kernels: parameter broadcast, gradient sinks feeding the
bucket logic, bucketed all-reduce = mean over ranks, fused AdamW on the averaged gradients -- against a single-process step on the
concatenated batch (mean of per-rank mean-loss gradients == gradient of the global mean loss)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    import simple_tad_amd as T
    torch.manual_seed(seed)
    return T.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, all_frames=4,
                               tubelet_size=2, num_classes=2, init_scale=1.0, drop_path_rate=0.0)


def _data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(8, 3, 4, 32, 32, generator=g), torch.randint(0, 2, (8,), generator=g)


def _worker(rank, world, port, q, backend="gloo"):
    """backend "gloo": both ranks on cuda:0 (one-GPU box); "nccl": one GPU per rank, RCCL over xGMI -- the production transport,
    through parallel.init_distributed_mode exactly as bench.py / a training script enters it"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if backend == "nccl" else 0), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import torch.nn.functional as F
    from simple_tad_amd import engine as E
    from simple_tad_amd.parallel import DataParallel, init_distributed_mode
    if backend == "nccl":
        ok, r, w, local = init_distributed_mode()
        assert ok and r == rank and w == world and local == rank and dist.get_backend() == "nccl"
        torch.cuda.set_device(local)
    else:
        dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
    try:
        m = _model(seed=100 + rank).cuda()           # different init per rank: the broadcast must make them identical
        dp = DataParallel(m, bucket_mb=0.25)          # several buckets
        opt = E.create_optimizer(dp, lr=1e-2, weight_decay=0.05, layer_decay=0.75)
        scaler = E.NativeScalerWithGradNormCount(dp)
        x, y = _data()
        xs, ys = x[rank * 4:(rank + 1) * 4].cuda(), y[rank * 4:(rank + 1) * 4].cuda()
        w0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}  # numpy: pickled by value through the queue
        m.train()
        dp.zero_grad()
        loss = F.cross_entropy(dp(xs), ys)
        loss.backward()
        dp.finish()
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        dp.zero_grad()
        norm = scaler(F.cross_entropy(dp(xs), ys), opt, parameters=list(m.parameters()))
        w1 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
        q.put((rank, len(dp.buckets), w0, grads, float(norm), w1))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process_step():
    _two_rank_step("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has one); runs on multi-GPU nodes")
def test_two_ranks_over_rccl_match_single_process_step():
    """the `backend == "nccl"` branch of init_distributed_mode and DataParallel's stream-ordering assumption (RCCL orders its kernels
    after the current stream), on the real transport"""
    _two_rank_step("nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_bench_self_launch_two_gpus_over_rccl():
    """`python bench.py --gpus 2` with no launcher: one rank-0 JSON line, n_gpus = the RCCL-reported world size"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "8"],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collective"]["backend"] == "nccl" and out["collective"]["world_size"] == 2
    assert out["config"]["global_batch"] == 16 and out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.parametrize("extra,gb,scaling", [([], 8, "weak"), (["--global-batch", "8"], 8, "strong"), (["--bucket-dtype", "bf16"], 8, "weak")])
def test_bench_two_ranks_over_gloo_on_one_gpu_reports_the_exchange(extra, gb, scaling):
    """The N > 1 path of bench.py end to end on ONE GPU (two ranks share it, gradients travel over gloo): one rank-0 JSON line with
    n_gpus = 2 and a `collective` object that says how the exchange went -- exposed all-reduce time, per-bucket time, per-rank step
    times, the Linear schedule in force -- so that the first run on a real 8-GPU node diagnoses itself.  Weak
    scaling (4 clips per rank) and the strong-scaling path (--global-batch)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TAD_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stderr[-3000:])
    out = json.loads(lines[0])
    c = out["collective"]
    assert out["n_gpus"] == 2 and c["backend"] == "gloo" and c["world_size"] == 2 and out["scaling"] == scaling
    assert out["config"]["global_batch"] == gb and out["config"]["per_gpu_batch"] == 4 and out["value"] > 0
    assert c["buckets"] == len(c["bucket_ms"]) == len(c["bucket_mbytes"]) >= 2 and c["steps"] == 2
    assert abs(sum(c["bucket_mbytes"]) * 2 ** 20 - c["allreduce_bytes_per_step"]) < 2 ** 20
    assert all(v is not None and v > 0 for v in c["bucket_ms"]) and c["exposed_ms"] > 0 and c["late_buckets_per_step"] == 0
    assert len(c["rank_ms_per_step"]["all"]) == 2 and c["rank_ms_per_step"]["min"] <= c["rank_ms_per_step"]["max"] <= out["ms_per_step"] * 1.05
    assert "per-tile" in c["linear_schedule"] and c["avg_in_collective"] is False
    assert c["bucket_dtype"] == ("bf16" if "bf16" in extra else "f32") and "version" in c["rccl"] or "version_error" in c["rccl"]
    assert c["allreduce_bytes_per_step"] == (2 if "bf16" in extra else 4) * (c["allreduce_bytes_per_step"] // (2 if "bf16" in extra else 4))


def _two_rank_step(backend):
    import torch.nn.functional as F
    from simple_tad_amd import engine as E
    from simple_tad_amd.parallel import DataParallel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}  # noqa: E731
    (_, nb0, w00, g0, n0, w10), (_, nb1, w01, g1, n1, w11) = [(r[0], r[1], t(r[2]), t(r[3]), r[4], t(r[5])) for r in res]
    assert nb0 >= 3 and nb0 == nb1
    for k in w00:
        assert torch.equal(w00[k], w01[k]), k          # broadcast: replicas start identical (rank 0's weights)
        assert torch.equal(g0[k], g1[k]) if k in g0 else True
        assert torch.equal(w10[k], w11[k]), k          # and stay identical after the step
    assert n0 == n1
    # single-process reference on the full batch with rank 0's initial weights
    m = _model(seed=100).cuda()
    m.load_state_dict(w00)
    dp = DataParallel(m)
    opt = E.create_optimizer(dp, lr=1e-2, weight_decay=0.05, layer_decay=0.75)
    scaler = E.NativeScalerWithGradNormCount(dp)
    x, y = _data()
    m.train()
    dp.zero_grad()
    F.cross_entropy(dp(x.cuda()), y.cuda()).backward()
    errs = {}
    for k, p in m.named_parameters():
        ref = p.grad.detach().cpu()
        errs[k] = ((g0[k] - ref).norm() / ref.norm().clamp_min(1e-12)).item()
    bad = {k: round(e, 6) for k, e in errs.items() if e >= 1e-5}
    assert not bad, bad   # the kernels are batch-invariant: mean of per-rank gradients == full-batch gradient to f32 rounding
    dp.zero_grad()
    norm = scaler(F.cross_entropy(dp(x.cuda()), y.cuda()), opt, parameters=list(m.parameters()))
    assert abs(float(norm) - n0) < 1e-5 * n0
    for k, v in m.state_dict().items():
        d = (w10[k] - w00[k]).norm()
        assert ((w10[k] - v.cpu()).norm() <= 1e-3 * d + 1e-7), k   # same update


# ------------------------------------------------------------------ RCCL through the C ABI (tad_rccl_*: the route of a host without torch)
def _rccl_rank(rank, world, idbytes, q):
    import ctypes as C
    from simple_tad_amd import _lib
    from simple_tad_amd._lib import TAD_BF16, TAD_F32, check
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(rank)
    lib = _lib.load()
    comm = C.c_void_p()
    check(lib.tad_rccl_init(idbytes, world, rank, C.byref(comm)), "tad_rccl_init")
    n = C.c_int()
    check(lib.tad_rccl_world_size(comm, C.byref(n)), "tad_rccl_world_size")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    parts = [torch.randn(1 << 20, generator=g) for _ in range(world)]          # every rank knows every rank's contribution
    x = parts[rank].cuda()
    check(lib.tad_rccl_allreduce(comm, x.data_ptr(), x.numel(), TAD_F32, 1, st), "tad_rccl_allreduce")      # average, in place
    xb = parts[rank].cuda().bfloat16()
    check(lib.tad_rccl_allreduce(comm, xb.data_ptr(), xb.numel(), TAD_BF16, 0, st), "tad_rccl_allreduce")   # sum, bf16
    w = (parts[0] if rank == 0 else torch.zeros(1 << 20)).cuda()
    check(lib.tad_rccl_broadcast(comm, w.data_ptr(), w.numel(), TAD_F32, 0, st), "tad_rccl_broadcast")
    torch.cuda.synchronize()
    mean = torch.stack(parts).mean(0)
    ssum = torch.stack([p.bfloat16().float() for p in parts]).sum(0)
    ok = (n.value == world and (x.cpu() - mean).abs().max().item() < 1e-6 and torch.equal(w.cpu(), parts[0])
          and (xb.float().cpu() - ssum).abs().max().item() <= 2.0 ** -7 * ssum.abs().max().item())
    assert lib.tad_rccl_allreduce(comm, x.data_ptr(), x.numel(), 7, 0, st) == -1  # bad dtype: refused on the host
    check(lib.tad_rccl_destroy(comm), "tad_rccl_destroy")
    if q is not None:
        q.put((rank, ok))
    return ok


def _rccl_unique_id():
    import ctypes as C
    from simple_tad_amd import _lib
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().tad_rccl_unique_id(buf), "tad_rccl_unique_id")
    return buf.raw


def test_rccl_c_abi_single_rank():
    """communicator of one rank on cuda:0: all-reduce (average / sum, f32 / bf16) and broadcast are identities, the calls are
    ordered on the caller's stream, errors come back as codes"""
    assert _rccl_rank(0, 1, _rccl_unique_id(), None)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_rccl_c_abi_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idb = _rccl_unique_id()
    procs = [ctx.Process(target=_rccl_rank, args=(r, 2, idb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _nccl_world1(q):
    """one rank over the REAL nccl (= RCCL) backend on cuda:0: process-group creation through init_distributed_mode, DataParallel's
    probe of ReduceOp.AVG and its bucket exchange (an identity with one rank) -- the RCCL code path that the two-rank tests cannot
    reach on a one-GPU box"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from simple_tad_amd.parallel import DataParallel
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        torch.cuda.set_device(0)
        m = _model(seed=3).cuda()
        dp = DataParallel(m, bucket_mb=0.25)
        g = torch.randn_like(dp.flat_grad)
        dp.flat_grad.copy_(g)
        works = [dp._exchange(b, async_op=True) for b in dp.buckets]
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        q.put((dist.get_backend(), bool(dp._avg_in_collective), torch.equal(dp.flat_grad, g), len(dp.buckets)))
    finally:
        dist.destroy_process_group()


def test_nccl_backend_single_rank_bucket_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1, args=(q,))
    p.start()
    backend, avg, same, nb = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and backend == "nccl" and same and nb >= 3
    print("ReduceOp.AVG inside the RCCL collective:", avg)
