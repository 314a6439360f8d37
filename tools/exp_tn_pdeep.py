#!/usr/bin/env python3
"""A/B of the deeper P-operand prefetch of gemm_tn (tad_linear_tuning("tn_pdeep")): 0 = two-stage ring (rounds 1-3), 1 = P ring of three
slots, P two reduction tiles ahead.  Interleaved rounds in one process; results must be bit-identical (same arithmetic, same order).
    python tools/exp_tn_pdeep.py [--rounds 7] [--iters 10] [--D 768] [--M 50176]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--M", type=int, default=50176)
ap.add_argument("--knob", default="tn_pdeep", help="tn_pdeep (round 4) or tn_w4 (round 5: the four-wave kernel of csrc/gemm_w4.hip)")
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
M, D, dev, bf = a.M, a.D, "cuda", (torch.float16 if a.dtype == "f16" else torch.bfloat16)
torch.manual_seed(0)


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


tot = [0.0, 0.0]
print(f"{'dW shape':28s} {a.knob + '=0 us':>12s} {a.knob + '=1 us':>12s}  bit-identical   ({a.dtype})")
try:
    for name, n, k in (("qkv  [3D, D]", 3 * D, D), ("proj [D, D]", D, D), ("fc1  [4D, D]", 4 * D, D), ("fc2  [D, 4D]", D, 4 * D)):
        dy, x = torch.randn(M, n, device=dev).to(bf), torch.randn(M, k, device=dev).to(bf)
        fn = lambda: K.linear_bwd_weight(dy, x, want_bias=True)  # noqa: E731
        outs = []
        for v in (0, 1):
            K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, a.knob: v})
            dW, db = fn()
            outs.append((dW.clone(), db.clone()))
            for _ in range(3):
                fn()
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        t = [[], []]
        for _ in range(a.rounds):
            for v in (0, 1):
                K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, a.knob: v})
                t[v].append(timeit(fn))
        m0, m1 = statistics.median(t[0]), statistics.median(t[1])
        tot[0] += m0
        tot[1] += m1
        print(f"{name:28s} {m0:12.1f} {m1:12.1f}  {same}   {2.0 * M * n * k / m0 / 1e6:6.0f} -> {2.0 * M * n * k / m1 / 1e6:6.0f} TFLOP/s", flush=True)
finally:
    K.linear_tuning(**K.LINEAR_TUNING_DEFAULTS)
print(f"{'sum':28s} {tot[0]:12.1f} {tot[1]:12.1f}")
