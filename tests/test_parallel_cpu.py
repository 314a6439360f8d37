"""world_size-2 test of the data-parallel gradient exchange over gloo (CPU): bucketed async all-reduce of a flat
gradient buffer must equal the mean of the per-rank gradients, replicas must start identical (broadcast), and
gradient accumulation must exchange only on the last micro-step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(16, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from simple_tad_amd.parallel import DataParallel, init_distributed_mode
    from simple_tad_amd import engine as E
    ok, r, w, _ = init_distributed_mode(backend="gloo")
    assert ok and r == rank and w == world
    model = _make_model(seed=100 + rank)           # different init per rank: broadcast must fix it
    dp = DataParallel(model, bucket_mb=0.004)       # tiny buckets -> several all-reduces
    assert len(dp.buckets) >= 3
    p0 = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(p0) for _ in range(world)]
    dist.all_gather(gathered, p0)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    # per-rank data (seed + rank, as run_class_finetuning.py:222)
    torch.manual_seed(rank)
    xs = [torch.randn(8, 16) for _ in range(2)]
    ys = [torch.randint(0, 3, (8,)) for _ in range(2)]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    scaler = E.NativeScalerWithGradNormCount(dp)
    crit = nn.CrossEntropyLoss()
    dp.zero_grad()
    # two micro-steps (update_freq=2): exchange only on the second
    loss = crit(dp(xs[0]), ys[0]) / 2
    assert scaler(loss, opt, parameters=list(model.parameters()), update_grad=False) is None
    assert not dp._works
    local_after_first = dp.flat_grad.clone()
    loss = crit(dp(xs[1]), ys[1]) / 2
    loss.backward()
    dp.require_sync = True
    # emulate the last micro-step by hand to capture the pre-step gradients
    # (the hook did not fire for this backward because require_sync was False) -> finish() handles nothing; so redo properly:
    dp.zero_grad()
    dp.require_sync = False
    (crit(dp(xs[0]), ys[0]) / 2).backward()
    dp.require_sync = True
    (crit(dp(xs[1]), ys[1]) / 2).backward()
    dp.finish()
    g_avg = dp.flat_grad.clone()
    # reference: mean over ranks of locally accumulated grads, computed with plain autograd + all_gather
    ref_model = _make_model(seed=100)  # rank-0 init == broadcast result
    ref_model.load_state_dict(model.state_dict())
    for x, y in zip(xs, ys):
        (crit(ref_model(x), y) / 2).backward()
    local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in reversed(list(ref_model.parameters()))])
    allg = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allg, local)
    mean = torch.stack(allg).mean(0)
    mine = torch.cat([p.grad.flatten() for p in reversed(list(model.parameters()))])
    err = (mine - mean).abs().max().item()
    same_views = all(p.grad.data_ptr() == dp.space.grad_view(p).data_ptr() for p in model.parameters())
    # optimizer.zero_grad(set_to_none=True) must not break the flat buffer
    opt.zero_grad(set_to_none=True)
    (crit(dp(xs[0]), ys[0])).backward()
    dp.finish()
    rehomed = all(p.grad.data_ptr() == dp.space.grad_view(p).data_ptr() for p in model.parameters())
    # C4: epoch meters averaged over the ranks in one fp64 all-reduce (utils.py:71-82); None entries are skipped
    avg = E.synchronize_meters({"loss": [1.0 + rank, 3.0 + rank], "grad_norm": [None, 2.0 * (rank + 1)], "lr": [0.5, 0.5]})
    mean_rank = (world - 1) / 2.0
    meters_ok = avg == {"loss": 2.0 + mean_rank, "grad_norm": 2.0 * (mean_rank + 1), "lr": 0.5}
    # overlap=False: every bucket is exchanged in finish() (the mode for models that use a parameter twice per backward)
    m2 = _make_model(seed=100)
    dp2 = DataParallel(m2, bucket_mb=0.004, overlap=False)
    dp2.zero_grad()
    crit(dp2(xs[0]), ys[0]).backward()
    assert not dp2._works
    pre = dp2.flat_grad.clone()
    dp2.finish()
    allp = [torch.zeros_like(pre) for _ in range(world)]
    dist.all_gather(allp, pre)
    late_ok = (dp2.flat_grad - torch.stack(allp).mean(0)).abs().max().item() < 1e-6
    q.put((rank, err, same_views, rehomed, float(local_after_first.abs().sum()) > 0 and meters_ok and late_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, err, same_views, rehomed, had_local in res:
        assert err < 1e-6, (rank, err)
        assert same_views and rehomed and had_local


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from simple_tad_amd.parallel import DataParallel, init_distributed_mode
    init_distributed_mode(backend="gloo")
    crit = nn.CrossEntropyLoss()
    torch.manual_seed(rank)
    x, y = torch.randn(8, 16), torch.randint(0, 3, (8,))
    grads = {}
    for fmt, overlap in (("f32", True), ("bf16", True), ("bf16", False)):
        dp = DataParallel(_make_model(seed=100), bucket_mb=0.004, bucket_dtype=fmt, overlap=overlap)
        assert (dp._stage16 is not None) == (fmt == "bf16") and len(dp.buckets) >= 3
        dp.enable_timing()
        dp.zero_grad()
        crit(dp(x), y).backward()
        dp.finish()
        grads[(fmt, overlap)] = dp.flat_grad.clone()
        assert dp.timing_summary()["bucket_dtype"] == fmt
        assert dp.exchange_bytes_per_step() == dp.flat_grad.numel() * (2 if fmt == "bf16" else 4)
    ref = grads[("f32", True)]
    # every rank holds the SAME averaged gradient (the collective's sum is what each of them widens back)
    same = [torch.zeros_like(ref) for _ in range(world)]
    dist.all_gather(same, grads[("bf16", True)])
    identical = all(torch.equal(same[0], t) for t in same)
    scale = ref.abs().max().item()
    err = max(((grads[k] - ref).abs().max().item() / scale) for k in (("bf16", True), ("bf16", False)))
    # per element: one rounding per rank (2^-9 of that rank's share) + the bf16 addition of the collective (2^-9 of the sum), i.e. at most
    # 2^-8 of the mean MAGNITUDE of the ranks' contributions (relative to the mean itself it is unbounded where the ranks cancel)
    dpl = DataParallel(_make_model(seed=100), bucket_mb=0.004)
    dpl.require_sync = False
    dpl.zero_grad()
    crit(dpl(x), y).backward()
    mags = [torch.zeros_like(ref) for _ in range(world)]
    dist.all_gather(mags, dpl.flat_grad.abs())
    mag = torch.stack(mags).mean(0)
    nz = mag > 0
    rel = ((grads[("bf16", True)] - ref).abs()[nz] / mag[nz]).max().item()
    q.put((rank, err, rel, identical, torch.equal(grads[("bf16", True)], grads[("bf16", False)])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bf16_bucket_exchange_matches_the_f32_exchange_gloo():
    """SURVEY 2.3 C1 "fp32 or bf16 flat buckets" (VERDICT r05 item 6): at world 2 the averaged gradients of the bf16 exchange agree with the
    f32 exchange to 2^-8 relative, overlapped and in finish() alike, and every rank ends with the same bits"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, err, rel, identical, same_both_routes in res:
        # err: largest deviation relative to the largest gradient element (the verdict's bound); rel: per element, relative to the mean magnitude
        # of the ranks' contributions -- 2^-8 with round-to-nearest additions, 2^-7 allowed here because gloo's bf16 sum is not pinned to that
        assert err <= 2.0 ** -8 and rel <= 2.0 ** -7, (rank, err, rel)
        assert err > 0, "the bf16 exchange must actually have rounded something"
        assert identical and same_both_routes


def test_bucket_dtype_is_validated():
    from simple_tad_amd.parallel import DataParallel
    with pytest.raises(ValueError, match="bucket_dtype"):
        DataParallel(_make_model(0), bucket_dtype="fp8")
    dp = DataParallel(_make_model(0), bucket_dtype="bf16")  # world 1: nothing to stage
    assert dp._stage16 is None and dp.exchange_bytes_per_step() == 2 * dp.flat_grad.numel()


def test_flat_space_layout_and_views():
    """flat.FlatSpace: reverse registration order, every tensor on a 4096-element boundary, gradients/params re-homed as views."""
    import torch
    from simple_tad_amd.flat import FlatSpace
    lin = torch.nn.Sequential(torch.nn.Linear(10, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in lin.parameters()]
    lin[0].weight.grad = torch.full_like(lin[0].weight, 2.0)
    sp = FlatSpace(list(lin.parameters()))
    assert [tuple(p.shape) for p in sp.params] == [(3,), (3, 7), (7,), (7, 10)]
    assert all(o % FlatSpace.ALIGN == 0 for o in sp.offset.values()) and sp.total == 4 * FlatSpace.ALIGN
    fg = sp.ensure_grads()
    assert lin[0].weight.grad.data_ptr() == fg.data_ptr() + sp.offset[id(lin[0].weight)] * 4
    assert float(fg.sum()) == 2.0 * 70  # an existing gradient is kept, padding stays zero
    fp = sp.adopt_params()
    assert sp.params_are_flat() and all(torch.equal(a, b) for a, b in zip(before, lin.parameters()))
    lin[1].bias.data.add_(1.0)
    assert torch.equal(sp.view(fp, lin[1].bias), before[3] + 1.0)  # the parameter IS the flat storage
    lin[0].weight.grad = None
    sp.rehome_grad(lin[0].weight)
    assert lin[0].weight.grad.data_ptr() == sp.grad_view(lin[0].weight).data_ptr() and float(fg.sum()) == 0.0
    lin[0].weight.data = lin[0].weight.data.clone()
    assert not sp.params_are_flat()


@pytest.mark.parametrize("n,world", [(10, 2), (11, 4), (7, 8), (1000, 8), (3, 2)])
@pytest.mark.parametrize("shuffle,drop_last", [(True, False), (False, False), (True, True)])
def test_shard_sampler_matches_torch_distributed_sampler(n, world, shuffle, drop_last):
    """row 15 (index sharding): identical indices to torch.utils.data.DistributedSampler, which the reference uses
    (run_class_finetuning.py:239-241), for every rank and epoch; together the ranks cover the clip set"""
    from torch.utils.data import DistributedSampler
    from simple_tad_amd.parallel import ShardSampler
    data = list(range(n))
    for epoch in (0, 1, 5):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=shuffle, seed=3, drop_last=drop_last)
            ref.set_epoch(epoch)
            ours = ShardSampler(n, num_replicas=world, rank=rank, shuffle=shuffle, seed=3, drop_last=drop_last)
            ours.set_epoch(epoch)
            a, b = list(ref), list(ours)
            assert a == b and len(ours) == len(ref) == len(b)
            seen += b
        if not drop_last:
            assert set(seen) == set(data)
