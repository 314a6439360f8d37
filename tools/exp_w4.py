#!/usr/bin/env python3
"""The four-wave 256 x 256 gemm_nt kernels (csrc/gemm_w4.hip: 128 x 128 outputs per wave, one wave per SIMD, hand-ordered K loop) against the
eight-wave ones, interleaved in ONE process at the Linear shapes of a ViT-B block (M = 50176): bit-for-bit equality of the results and
median times.   python tools/exp_w4.py [--rounds 7] [--iters 10] [--dtype bf16|f16]
variant 0 = the planned launch (whole rounds + tail), 1 = 8-wave 256 x 256 for the whole problem, 7 = 4-wave 256 x 256 for the whole problem."""
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
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--M", type=int, default=50176)
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--variants", default="0,1,7")
a = ap.parse_args()
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
M, D, dev = a.M, a.D, "cuda"
variants = [int(v) for v in a.variants.split(",")]


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dt)


x_d, x_3d, x_4d = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
res = torch.randn(M, D, device=dev)
hpre = rnd(M, 4 * D)
b_d, b_3d, b_4d = (torch.randn(n, device=dev) for n in (D, 3 * D, 4 * D))
W = {n: rnd(*s, scale=0.02) for n, s in {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D), "fc2T": (4 * D, D), "fc1T": (D, 4 * D),
                                         "qkvT": (D, 3 * D)}.items()}
cases = [
    ("qkv fwd plain", 2.0 * M * 3 * D * D, lambda: K.linear_fwd(x_d, W["qkv"], b_3d)[0]),
    ("qkv fwd (q prescale)", 2.0 * M * 3 * D * D, lambda: K.linear_fwd_qkv(x_d, W["qkv"], b_d, b_d, q_prescale=0.18)),
    ("proj fwd +res f32", 2.0 * M * D * D, lambda: K.linear_fwd(x_d, W["proj"], b_d, out_dtype=torch.float32, epilogue=2, residual=res)[0]),
    ("fc1 fwd gelu +preact", 2.0 * M * 4 * D * D, lambda: torch.cat(K.linear_fwd(x_d, W["fc1"], b_4d, epilogue=1, want_preact=True))),
    ("fc2 fwd +res f32", 2.0 * M * 4 * D * D, lambda: K.linear_fwd(x_4d, W["fc2"], b_d, out_dtype=torch.float32, epilogue=2, residual=res)[0]),
    ("dX fc2 (dgelu) 16-bit", 2.0 * M * 4 * D * D, lambda: K.linear_bwd_input(x_d, W["fc2T"], gelu_preact=hpre)),
    ("dX fc1 f32", 2.0 * M * 4 * D * D, lambda: K.linear_bwd_input(x_4d, W["fc1T"], out_dtype=torch.float32)),
    ("dX proj 16-bit", 2.0 * M * D * D, lambda: K.linear_bwd_input(x_d, W["proj"])),
    ("dX qkv f32", 2.0 * M * 3 * D * D, lambda: K.linear_bwd_input(x_3d, W["qkvT"], out_dtype=torch.float32)),
]


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


print(f"M={M} D={D} {a.dtype}; us (TF/s) per variant, medians of {a.rounds} interleaved rounds x {a.iters} launches; '=' results bit-identical to variant {variants[0]}")
print(f"{'case':24s}" + "".join(f"{'v' + str(v):>18s}" for v in variants))
tot = {v: 0.0 for v in variants}
try:
    for label, flops, fn in cases:
        outs, t = {}, {v: [] for v in variants}
        for v in variants:
            K.linear_tuning(variant=v)
            outs[v] = fn().clone()
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for v in variants:
                K.linear_tuning(variant=v)
                t[v].append(timeit(fn))
        line = f"{label:24s}"
        for v in variants:
            med = statistics.median(t[v])
            tot[v] += med
            same = "=" if torch.equal(outs[v], outs[variants[0]]) else "!"
            line += f"{med:9.1f} ({flops / med / 1e6:5.0f}){same}"
        print(line, flush=True)
finally:
    K.linear_tuning(variant=0)
print(f"{'sum':24s}" + "".join(f"{tot[v]:9.1f}{'':9s}" for v in variants))
