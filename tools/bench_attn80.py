#!/usr/bin/env python3
"""head_dim 80 (vit_huge: 16 heads x 80) attention at the fine-tune shape, 16-bit MFMA kernels against the exact-f32 kernels that
served this head dim until round 3 (interleaved rounds, one process):  python tools/bench_attn80.py [--B 32] [--rounds 5]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--H", type=int, default=16)
ap.add_argument("--N", type=int, default=1568)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
B, H, N, d = a.B, a.H, a.N, 80
scale = d ** -0.5
torch.manual_seed(0)
q32 = torch.randn(B * N, 3 * H * d, device="cuda")
q16 = q32.to(torch.bfloat16)
q16[:, :H * d] = (q32[:, :H * d] * K.q_prescale_of(scale)).to(torch.bfloat16)
do32 = torch.randn(B * N, H * d, device="cuda")
do16 = do32.to(torch.bfloat16)
o16, lse16, lo16 = K.attn_fwd(q16, B, N, H, scale, want_lo=True, q_prescaled=True, d=d)
o32, lse32 = K.attn_fwd_f32(q32, B, N, H, scale, want_lse=True, d=d)
print("max |o16 - o32| =", float((o16.float() - o32).abs().max()))
fns = {
    "fwd 16-bit": lambda: K.attn_fwd(q16, B, N, H, scale, want_lo=True, q_prescaled=True, d=d),
    "bwd 16-bit": lambda: K.attn_bwd(q16, o16, do16, lse16, B, N, H, scale, out_lo=lo16, q_prescaled=True, d=d),
    "fwd f32": lambda: K.attn_fwd_f32(q32, B, N, H, scale, want_lse=True, d=d),
    "bwd f32": lambda: K.attn_bwd_f32(q32, o32, do32, lse32, B, N, H, scale, d=d),
}


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


for fn in fns.values():
    fn()
res = {k: [] for k in fns}
for _ in range(a.rounds):
    for k, fn in fns.items():
        res[k].append(timeit(fn))
fl = 4.0 * B * H * N * N * d
for k, v in res.items():
    us = statistics.median(v)
    print(f"{k:12s} {us:9.1f} us   {(fl if k.startswith('fwd') else 2.5 * fl) / us / 1e6:7.1f} TFLOP/s (fwd 4BHNNd, bwd 10BHNNd)")
print("speed-up fwd %.2fx  bwd %.2fx" % (statistics.median(res["fwd f32"]) / statistics.median(res["fwd 16-bit"]),
                                        statistics.median(res["bwd f32"]) / statistics.median(res["bwd 16-bit"])))
