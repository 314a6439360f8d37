#!/usr/bin/env python3
"""Calibration, not a product path: the vendor library's bf16 GEMMs (torch.nn.functional.linear / torch.matmul -> hipBLASLt / rocBLAS) at the
Linear shapes of a ViT-B block (M = 50176 rows, random data) beside this library's hand-written kernels, interleaved rounds in one process.
What the chip's own tuned assembly reaches on these shapes is the practical ceiling the roofline fractions in DESIGN.md should be read against.

    python tools/exp_blas_compare.py [--rounds 7] [--iters 10]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev, bf, D, M = "cuda", torch.bfloat16, 768, 50176


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def ab(name, flop, fns):
    for f in fns.values():
        for _ in range(3):
            f()
    t = {k: [] for k in fns}
    for _ in range(a.rounds):
        for k, f in fns.items():
            t[k].append(timed(f, a.iters))
    line = f"{name:34s}"
    for k in fns:
        med = float(np.median(t[k]))
        line += f" | {k}: {med:7.1f} us {flop / med / 1e6:6.0f} TF"
    print(line, flush=True)
    return {k: float(np.median(v)) for k, v in t.items()}


tot = {}
print("== y = x W^T (gemm_nt): plain bf16 output, with bias")
for name, n, k in [("qkv   N2304 K768", 3 * D, D), ("fc1   N3072 K768", 4 * D, D), ("fc2   N768  K3072", D, 4 * D), ("proj  N768  K768", D, D),
                   ("dXqkv N768  K2304", D, 3 * D)]:
    x = torch.randn(M, k, device=dev).to(bf)
    w = (torch.randn(n, k, device=dev) * 0.02).to(bf)
    b32 = torch.randn(n, device=dev)
    bb = b32.to(bf)
    r = ab(name, 2.0 * M * n * k, {"vendor linear+bias": lambda: torch.nn.functional.linear(x, w, bb), "vendor matmul": lambda: torch.matmul(x, w.t()),
                                   "tad linear_fwd": lambda: K.linear_fwd(x, w, b32)})
    for kk, v in r.items():
        tot[kk] = tot.get(kk, 0.0) + v
    del x, w
print("== dW = dy^T x (gemm_tn), f32 output here / bf16->f32 there")
for name, n, k in [("dW qkv  2304x768", 3 * D, D), ("dW fc1  3072x768", 4 * D, D), ("dW fc2  768x3072", D, 4 * D), ("dW proj 768x768", D, D)]:
    dy = torch.randn(M, n, device=dev).to(bf)
    x = torch.randn(M, k, device=dev).to(bf)
    r = ab(name, 2.0 * M * n * k, {"vendor matmul (bf16 out)": lambda: torch.matmul(dy.t(), x), "tad linear_bwd_weight (+db, f32)": lambda: K.linear_bwd_weight(dy, x, want_bias=True)})
    for kk, v in r.items():
        tot[kk] = tot.get(kk, 0.0) + v
    del dy, x
print("sums (us):", "  ".join(f"{k}: {v:.1f}" for k, v in tot.items()))
print("torch", torch.__version__, "| preferred BLAS:", getattr(torch.backends.cuda, "preferred_blas_library", lambda: "n/a")())
