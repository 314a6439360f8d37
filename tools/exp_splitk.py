#!/usr/bin/env python3
"""A/B of the split-K tail of the Linear GEMMs (tad_linear_tuning("splitk_tail")): 0 = off (round-3 plans), 1 = cost model, 2 = forced.
Interleaved rounds in one process, random data, rows = 32 clips x 1568 tokens.   python tools/exp_splitk.py [--rounds 7] [--iters 10]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--M", type=int, default=32 * 1568)
ap.add_argument("--only", default="", help="substring filter on the shape names")
a = ap.parse_args()
dev, bf, M = "cuda", torch.bfloat16, a.M
torch.manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(bf)


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


shapes = [("ViT-B proj fwd   N 768 K 768  +res", 768, 768, "res"), ("ViT-B fc2 fwd    N 768 K 3072 +res", 768, 3072, "res"),
          ("ViT-B dX(fc1)    N 768 K 3072 bf16", 768, 3072, "plain"), ("ViT-B dX(qkv)    N 768 K 2304 bf16", 768, 2304, "plain"),
          ("ViT-B dX(proj)   N 768 K 768  bf16", 768, 768, "plain"),
          ("ViT-L proj fwd   N 1024 K 1024 +res", 1024, 1024, "res"), ("ViT-L fc2 fwd    N 1024 K 4096 +res", 1024, 4096, "res"),
          ("ViT-L dX(fc1)    N 1024 K 4096 bf16", 1024, 4096, "plain"), ("ViT-L dX(qkv)    N 1024 K 3072 bf16", 1024, 3072, "plain"),
          ("ViT-S fc2 fwd    N 384 K 1536 +res", 384, 1536, "res"), ("ViT-S dX(qkv)    N 384 K 1152 bf16", 384, 1152, "plain")]
shapes = [sh for sh in shapes if a.only in sh[0]]
print(f"{'shape':40s} {'off us':>9s} {'auto us':>9s} {'forced us':>10s}")
tot = [0.0, 0.0, 0.0]
try:
    for name, N, Kd, mode in shapes:
        x, w, b = rnd(M, Kd), rnd(N, Kd, scale=0.02), torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if mode == "res" else None
        fn = (lambda: K.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=res)) if mode == "res" else (lambda: K.linear_fwd(x, w, b))
        t = {0: [], 1: [], 2: []}
        for v in (0, 1, 2):
            K.linear_tuning(splitk_tail=v)
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for v in (0, 1, 2):
                K.linear_tuning(splitk_tail=v)
                t[v].append(timeit(fn))
        med = [statistics.median(t[v]) for v in (0, 1, 2)]
        for i in range(3):
            tot[i] += med[i] if name.startswith("ViT-B") else 0.0
        print(f"{name:40s} {med[0]:9.1f} {med[1]:9.1f} {med[2]:10.1f}")
        del x, w, res
finally:
    K.linear_tuning(**K.LINEAR_TUNING_DEFAULTS)
print(f"{'ViT-B sum (one block)':40s} {tot[0]:9.1f} {tot[1]:9.1f} {tot[2]:10.1f}   auto - off: {(tot[1] - tot[0]) * 12e-3:+.2f} ms per step")
