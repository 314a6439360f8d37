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
