"""Data-parallel gradient exchange for the fine-tuning step: one process per GPU, model replicated,
clip batch sharded, ONE exchange per optimizer step = all-reduce(mean) of the gradients.

Replaces the reference's ``torch.nn.parallel.DistributedDataParallel`` wrap
(run_class_finetuning.py:446-448, run_frame_finetuning.py:539-541) and ``utils.init_distributed_mode``
(utils.py:283-333).  Transport is RCCL over xGMI through ``torch.distributed`` (backend "nccl" on ROCm);
on CPU the same code runs over gloo (tests).

Design for xGMI (point-to-point links, ring collectives are per-link bound):
  * all gradients live in ONE flat f32 buffer laid out in reverse registration order (head first, patch-embed
    last), i.e. roughly the order in which backward produces them; ``p.grad`` are views into it;
  * the buffer is cut into a few large buckets (default 64 MiB: one to two transformer blocks) so each
    all-reduce is bandwidth- rather than latency-bound;
  * a bucket's all-reduce is launched asynchronously from a post-accumulate-grad hook as soon as its last
    gradient has been written, so communication overlaps the rest of backward; ``finish()`` waits for all of
    them before the optimizer step.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .flat import FlatSpace


def init_distributed_mode(backend: Optional[str] = None):
    """utils.init_distributed_mode (utils.py:283-333): read RANK / WORLD_SIZE / LOCAL_RANK (torchrun / srun / OMPI),
    bind the GPU, create the process group.  Returns (distributed, rank, world_size, local_rank)."""
    env = os.environ
    if "OMPI_COMM_WORLD_RANK" in env and "RANK" not in env:
        env["RANK"] = env["OMPI_COMM_WORLD_RANK"]
        env["WORLD_SIZE"] = env["OMPI_COMM_WORLD_SIZE"]
        env["LOCAL_RANK"] = env["OMPI_COMM_WORLD_LOCAL_RANK"]
    if "RANK" not in env or "WORLD_SIZE" not in env or int(env["WORLD_SIZE"]) <= 1:
        return False, 0, 1, int(env.get("LOCAL_RANK", 0))
    rank, world, local = int(env["RANK"]), int(env["WORLD_SIZE"]), int(env.get("LOCAL_RANK", 0))
    if backend is None:
        # TAD_DIST_BACKEND=gloo: debugging aid -- run the N > 1 path with several ranks on fewer GPUs (RCCL needs one GPU per rank)
        backend = env.get("TAD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    dist.barrier()
    return True, rank, world, local


class ShardSampler:
    """Index sharding of the clip set across ranks: the role `torch.utils.data.DistributedSampler(dataset, num_replicas,
    rank, shuffle=True, seed)` plays in the reference (run_class_finetuning.py:239-241, `set_epoch` at :502; shuffle=False
    for validation, :248-254).  Same published algorithm, hence the same indices: permutation of range(n) from a
    torch.Generator seeded `seed + epoch`, padded by wrap-around to a multiple of the world size (or truncated with
    drop_last), rank r takes positions r, r + world, r + 2 world, ...  Usable as the `sampler=` of a DataLoader."""

    def __init__(self, n: int, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        self.n, self.num_replicas, self.rank, self.shuffle, self.seed, self.drop_last = int(n), num_replicas, rank, shuffle, seed, drop_last
        self.epoch = 0
        if drop_last and self.n % num_replicas:
            self.num_samples = (self.n - num_replicas + num_replicas - 1) // num_replicas  # = n // world
        else:
            self.num_samples = (self.n + num_replicas - 1) // num_replicas
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad > 0:
                idx += (idx * ((pad + len(idx) - 1) // max(len(idx), 1) + 1))[:pad] if pad > len(idx) else idx[:pad]
        else:
            idx = idx[:self.total_size]
        return iter(idx[self.rank:self.total_size:self.num_replicas])


class DataParallel(nn.Module):
    """Replicated-model data parallelism with bucketed, overlapped gradient all-reduce."""

    def __init__(self, module: nn.Module, bucket_mb: float = 64.0, process_group=None, broadcast: bool = True, overlap: bool = True,
                 reduce_avg: Optional[bool] = None, tail_mb: Optional[float] = 16.0, linear_schedule: Optional[str] = None,
                 bucket_dtype: str = "f32"):
        """``tail_mb``: the buckets that complete LAST (block 0 and the patch embedding: backward produces gradients head first) are cut
        to this size.  Every earlier bucket's all-reduce runs beside the rest of backward; the last one has nothing left to hide behind,
        so its size IS the exposed time of the exchange (a ring all-reduce is per-link bound on xGMI: ~64 MiB takes ~1.1 ms at 8
        ranks, 16 MiB a quarter of that).  None: uniform buckets.
        ``overlap=False``: exchange every bucket in ``finish()`` after backward instead of as soon as its last gradient lands --
        required for models that use a parameter more than once per backward (weight tying, a module called twice per forward):
        the overlapped exchange announces a parameter after its FIRST gradient write and raises if it sees a second one.
        ``bucket_dtype="bf16"`` (SURVEY 2.3 C1 "fp32 or bf16 flat buckets"; off by default): every bucket travels as bfloat16 -- the
        rank's f32 gradients are divided by the world size, rounded ONCE to bf16 into a staging buffer, summed by the collective in bf16,
        and widened back into the f32 flat buffer -- half the bytes on a per-link-bound xGMI ring (ViT-B: 172 instead of 345 MB per
        step) for a relative error of the averaged gradient of at most ~2^-8 (one rounding per rank + the collective's bf16 additions;
        tests/test_parallel_cpu.py bounds it at world 2).  Bucket sizes (``bucket_mb`` / ``tail_mb``) keep counting f32 bytes, so the
        bucket boundaries, and with them the overlap pattern, are the same in both formats."""
        super().__init__()
        if bucket_dtype not in ("f32", "bf16"):
            raise ValueError(f"bucket_dtype must be 'f32' or 'bf16', got {bucket_dtype!r}")
        self.bucket_dtype = bucket_dtype
        self.module = module
        self.pg = process_group
        self.overlap = bool(overlap)
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        assert params, "no trainable parameters"
        dev, dt = params[0].device, params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in params), "parameters must share device and dtype"
        if broadcast and self.world > 1:  # C2: replicas start identical
            flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
            dist.broadcast(flat, src=0, group=self.pg)
            off = 0
            with torch.no_grad():
                for p in module.parameters():
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
        self.linear_schedule = None
        self._tuning = None
        if self.world > 1 and dev.type == "cuda":
            # The persistent Linear kernels expect one workgroup per CU with a fixed tile list each (132-147 KB of LDS, every VGPR).
            # While an RCCL collective runs beside the backward pass it holds some CUs, the workgroups meant for them start a
            # whole kernel late and their tile lists double that kernel's time; the per-tile grid lets the dispatcher balance
            # instead.  (Costs the next-tile prefetch and the peeled last K-tile: 0.2 % of a single-GPU step on the round-2 build.)
            # That choice was made with two gloo ranks sharing one GPU, never with RCCL on separate GPUs: `linear_schedule="persistent"`
            # (bench.py --linear-schedule) keeps the single-GPU schedule so that the first multi-GPU run can report both.
            # Split-K tails stay ON in their deferred three-launch form, which is ordered by kernel boundaries and needs no
            # co-residency; only the in-launch combine (splitk_defer=0) does, so it is pinned off here.
            self.set_linear_schedule(linear_schedule or "per-tile")
        # flat gradient buffer (layout shared with the fused optimizer: flat.FlatSpace)
        self.space = FlatSpace(params)
        self.flat_grad = self.space.ensure_grads()
        order = self.space.params
        offs = [self.space.offset[id(p)] for p in order]
        # buckets = contiguous ranges of the flat buffer
        limit = int(bucket_mb * (1 << 20) / self.flat_grad.element_size())
        tail = int(tail_mb * (1 << 20) / self.flat_grad.element_size()) if tail_mb else 0
        total = self.space.total
        self.buckets: List[dict] = []
        start, count, was_tail = 0, 0, False
        for i, (p, o) in enumerate(zip(order, offs)):
            end = o + FlatSpace.padded(p)
            # the last 2 x tail elements of the buffer go out in buckets of `tail` elements: close the running bucket where that region starts
            in_tail = bool(tail) and tail < limit and total - o <= 2 * tail
            if in_tail and not was_tail and o > start:
                self.buckets.append({"lo": start, "hi": o, "n": count, "ready": 0})
                start, count = o, 0
            was_tail = in_tail
            # a parameter marked as never receiving a gradient (a learnable pos_embed that is added detached) stays in the flat
            # layout with a zero gradient but is not waited for: its bucket is announced by the others
            count += 0 if getattr(p, "_tad_never_grad", False) else 1
            if end - start >= (tail if in_tail else limit) or i == len(order) - 1:
                self.buckets.append({"lo": start, "hi": end, "n": count, "ready": 0})
                start, count = end, 0
        self._bucket_of = {}
        bi = 0
        for p, o in zip(order, offs):
            while o >= self.buckets[bi]["hi"]:
                bi += 1
            self._bucket_of[id(p)] = bi
        # gloo with GPU tensors (debugging / single-GPU tests of the N > 1 path) stages through host memory and was observed to
        # read the bucket before the kernels queued on the current stream had written it: drain the stream first.  RCCL ("nccl")
        # orders its kernels after the current stream itself.
        self._drain_first = bool(dist.is_initialized() and dist.get_backend(self.pg) == "gloo" and self.flat_grad.is_cuda)
        # Mean over the ranks.  Default on every backend: divide the bucket by the world size, then all-reduce(SUM) -- the path the
        # world-size-2 / 4 tests execute.  ``reduce_avg=True`` (or TAD_DP_AVG=1) takes the mean inside the collective instead
        # (ncclAvg: one pass over each bucket less); it is opt-in until a multi-GPU RCCL run has exercised it, and the one-element
        # probe below is a collective every rank must pass: a failure RAISES on every rank instead of being swallowed (a collective
        # that fails on one rank leaves the communicator aborted or out of step, which would only surface later as a hang).
        if reduce_avg is None:
            reduce_avg = os.environ.get("TAD_DP_AVG", "0") == "1"
        self._avg_in_collective = False
        if reduce_avg and self.world > 1:
            if not (dist.get_backend(self.pg) == "nccl" and self.flat_grad.is_cuda):
                raise ValueError("DataParallel(reduce_avg=True) needs the nccl (RCCL) backend with GPU gradients; gloo has no AVG")
            probe = torch.ones(1, dtype=torch.float32, device=self.flat_grad.device)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.pg)
            if abs(float(probe.item()) - 1.0) > 1e-6:
                raise RuntimeError(f"DataParallel: ReduceOp.AVG of ones returned {float(probe.item())!r}")
            self._avg_in_collective = True
        if self.bucket_dtype == "bf16" and self._avg_in_collective:
            raise ValueError("DataParallel: bucket_dtype='bf16' divides before it rounds; it does not combine with reduce_avg=True")
        # staging buffer of the 16-bit exchange: same offsets as the flat f32 gradient buffer
        self._stage16 = torch.empty_like(self.flat_grad, dtype=torch.bfloat16) if (self.bucket_dtype == "bf16" and self.world > 1) else None
        self._announced = set()
        self._works = []
        self._timing = None  # enable_timing(): per-step records of the exchange
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in order]
        # A second gradient for an already announced parameter must not slip through: the kernel-side route is caught in _on_sink,
        # the autograd route here.  Tensor hooks run BEFORE the accumulation and receive None when a Function returned no gradient
        # for the parameter (the kernel wrote it through the sink), a tensor when autograd is about to add a real one.
        self._hooks += [p.register_hook(self._real_grad_guard(p)) for p in order]
        self.space.install_sinks(self._on_sink)  # GPU: dW GEMMs accumulate in place and announce the parameter themselves
        self.require_sync = True

    _TWICE = ("DataParallel: a parameter received a second gradient in one backward pass after its bucket had been announced (weight "
              "tying / a module used twice per forward); build DataParallel with overlap=False for such models")

    def _guarding(self) -> bool:
        return self.world > 1 and self.require_sync and self.overlap

    def _real_grad_guard(self, p):
        def hook(g):
            if g is not None and self._guarding() and id(p) in self._announced:
                raise RuntimeError(self._TWICE)
            return None
        return hook

    def _on_sink(self, p):
        """called by the kernel-side gradient sinks (ops.linear_dw / layernorm_bwd_sunk) right after an in-place write into p.grad"""
        if self._guarding() and id(p) in self._announced:
            raise RuntimeError(self._TWICE)  # (either route came first: a sink write or an autograd accumulation)
        self._on_grad(p)

    # -- hook: runs on the autograd thread right after p.grad has been accumulated
    def _on_grad(self, p):
        self.space.rehome_grad(p)  # someone called optimizer.zero_grad(set_to_none=True): autograd allocated a fresh tensor
        if self.world == 1 or not self.require_sync or not self.overlap:
            return
        # A parameter can be announced twice in one backward pass: by the kernel-side gradient sink (ops.linear_dw /
        # layernorm_bwd_sunk, right after the in-place write) and again by autograd's post-accumulate hook, which torch also runs
        # for inputs whose Function.backward returned None.  Count it once, or a bucket would be exchanged before its last
        # gradient has been written.
        if id(p) in self._announced:
            return
        self._announced.add(id(p))
        b = self.buckets[self._bucket_of[id(p)]]
        b["ready"] += 1
        if b["ready"] == b["n"]:
            b["ready"] = 0
            t0 = self._mark()
            self._works.append((self._exchange(b, async_op=True), self._bucket_of[id(p)], t0))

    def _exchange(self, b, async_op):
        """mean over the ranks of one bucket, in place (f32 buckets) or through the bf16 staging buffer (`_complete` widens it back)"""
        view = self.flat_grad[b["lo"]:b["hi"]]
        if self._avg_in_collective:
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=async_op)
        view.div_(self.world)
        if self._stage16 is not None:
            view = self._stage16[b["lo"]:b["hi"]].copy_(view)  # the ONE rounding of this rank's contribution
        if self._drain_first:
            torch.cuda.current_stream().synchronize()
        w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        if not async_op:
            self._complete(b)
        return w

    def _complete(self, b):
        """behind a finished bucket exchange: the bf16 sum goes back into the f32 gradient buffer (nothing to do for f32 buckets)"""
        if self._stage16 is not None:
            self.flat_grad[b["lo"]:b["hi"]].copy_(self._stage16[b["lo"]:b["hi"]])

    def exchange_bytes_per_step(self) -> int:
        return int(self.flat_grad.numel()) * (2 if self.bucket_dtype == "bf16" else self.flat_grad.element_size())

    def set_linear_schedule(self, schedule: str):
        """'per-tile' (one workgroup per tile, the dispatcher balances around CUs an RCCL kernel holds) or 'persistent' (one workgroup
        per CU walks a tile list: the single-GPU schedule).  Process-wide knobs of the library (tad_linear_tuning): `close()` puts the
        single-GPU defaults back."""
        from .tuning import TuningScope
        if schedule not in ("per-tile", "persistent"):
            raise ValueError(f"linear_schedule must be 'per-tile' or 'persistent', got {schedule!r}")
        if getattr(self, "_tuning", None) is not None:
            self._tuning.close()
        # (lock=False: this wrapper keeps its plan for as long as it lives; a scope that held the process-wide lock that long would block
        #  every other thread's scoped change)
        self._tuning = TuningScope(persistent=int(schedule == "persistent"), splitk_defer=1, lock=False).__enter__()
        self.linear_schedule = schedule

    def close(self):
        """undo the process-wide Linear scheduling knobs this wrapper set: whatever was in force before it was built comes back (a later
        single-GPU model in the same process does not inherit this wrapper's plan)"""
        if getattr(self, "_tuning", None) is not None:
            self._tuning.close()
            self._tuning = None
        self.linear_schedule = None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # -- self-diagnosis of the exchange (bench.py `collective`): how long backward's tail waits for the collectives
    def enable_timing(self, on: bool = True):
        """Record, per step, when each bucket's all-reduce was launched and when the compute stream could proceed past it, and how
        long ``finish()`` kept the compute stream waiting (the EXPOSED part of the exchange; everything else ran beside backward).
        GPU tensors: HIP events on the compute stream (``work.wait()`` makes that stream wait for the collective's stream); CPU
        tensors (gloo tests): host clocks.  ``timing_summary()`` aggregates; the records cost two events per bucket and step."""
        self._timing = {"steps": [], "cur": None} if on else None

    def _mark(self):
        if self._timing is None:
            return None
        if self.flat_grad.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            return e
        import time
        return time.perf_counter()

    @staticmethod
    def _elapsed_ms(a, b):
        return a.elapsed_time(b) if isinstance(a, torch.cuda.Event) else (b - a) * 1e3

    def timing_summary(self):
        """{exposed_ms (mean per step), bucket_ms [per bucket: launch -> compute stream past it, mean], late_buckets_per_step
        (buckets exchanged synchronously in finish() because a gradient never arrived), steps}"""
        if self._timing is None or not self._timing["steps"]:
            return None
        if self.flat_grad.is_cuda:
            torch.cuda.synchronize()
        steps = self._timing["steps"]
        nb = len(self.buckets)
        per_bucket = [[] for _ in range(nb)]
        exposed, late = [], []
        for st in steps:
            exposed.append(self._elapsed_ms(st["finish"][0], st["finish"][1]))
            late.append(st["late"])
            for bi, t0, t1 in st["buckets"]:
                per_bucket[bi].append(self._elapsed_ms(t0, t1))
        mean = lambda v: (sum(v) / len(v)) if v else None  # noqa: E731
        return {"steps": len(steps), "exposed_ms": round(mean(exposed), 4), "exposed_ms_max": round(max(exposed), 4),
                "bucket_ms": [None if not v else round(mean(v), 4) for v in per_bucket],
                # (bytes on the wire: two per element with bf16 buckets)
                "bucket_mbytes": [round((b["hi"] - b["lo"]) * (2 if self.bucket_dtype == "bf16" else self.flat_grad.element_size()) / 2 ** 20, 1) for b in self.buckets],
                "late_buckets_per_step": round(mean(late), 2), "overlap": self.overlap, "avg_in_collective": self._avg_in_collective,
                "bucket_dtype": self.bucket_dtype}

    def finish(self):
        """Wait for every in-flight bucket (call after backward, before the optimizer step)."""
        rec = {"buckets": [], "late": 0, "finish": None} if self._timing is not None else None
        t_in = self._mark()
        for w, bi, t0 in self._works:
            w.wait()
            self._complete(self.buckets[bi])
            if rec is not None:
                rec["buckets"].append((bi, t0, self._mark()))
        self._works.clear()
        self._announced.clear()
        if self.world == 1 or not self.require_sync:
            return
        for i, b in enumerate(self.buckets):  # a parameter that received no gradient this step leaves its bucket incomplete
            if b["n"] == 0 and self.overlap:
                continue  # only never-grad parameters: zeros on every rank
            if b["ready"] or not self.overlap:
                b["ready"] = 0
                t0 = self._mark()
                self._exchange(b, async_op=False)
                if rec is not None:
                    rec["buckets"].append((i, t0, self._mark()))
                    rec["late"] += 1
        if rec is not None:
            rec["finish"] = (t_in, self._mark())
            self._timing["steps"].append(rec)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are views into the flat buffer and must stay allocated: zero in place.  Difference from the reference's
        ``optimizer.zero_grad()`` (set_to_none): a parameter that receives no gradient in a step keeps a ZERO gradient here, so AdamW
        still decays it and advances its moments, where torch skips a ``None`` gradient.  Every parameter of the hot-path models
        receives a gradient in every step, so the trajectories agree (tests/test_optim_gpu.py covers the None case of FusedAdamW)."""
        self.flat_grad.zero_()
        self._announced.clear()

    def grad_sumsq_buffer(self):
        return self.flat_grad
