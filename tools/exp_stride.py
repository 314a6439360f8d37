#!/usr/bin/env python3
"""Experiment: does the row stride of the A operand limit gemm_nt's staging rate?  Same tile traffic, two shapes:
fc2 (M=50176, K=3072: A rows 6 KB apart) vs a 4x taller K=768 problem (A rows 1.5 KB apart).  Run under TAD_GEMM_DEBUG=6
(no MFMA, no epilogue: staging + fragment reads only) and =4 (no epilogue)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K
from tools.bench_kernels import timeit
dev, bf = "cuda", torch.bfloat16
K.linear_tuning(split_tail=0)
for name, M, N, Kd in [("fc2  M=50176  K=3072", 50176, 768, 3072), ("tall M=200704 K=768 ", 200704, 768, 768),
                       ("fc2 as a view of a [M, 12288] tensor (24 KB stride)", 50176, 768, 3072)]:
    if "view" in name:
        big = torch.randn(M, 4 * Kd, device=dev).to(bf)
        x = big[:, :Kd]  # non-contiguous view: not accepted by the wrapper -> skip unless strides are supported
        continue
    x = torch.randn(M, Kd, device=dev).to(bf)
    w = (torch.randn(N, Kd, device=dev) * 0.02).to(bf)
    ms = timeit(lambda: K.linear_fwd(x, w, None), 20)
    print(f"{name}: {ms * 1000:8.1f} us  (debug={os.environ.get('TAD_GEMM_DEBUG', '0')})", flush=True)

print("--- gemm_tn (dW): same tile traffic, operand row strides 1.5 KB + 6 KB vs 1.5 KB + 1.5 KB")
for name, M, N, Kd in [("dWfc2 M=50176  N=768 K=3072", 50176, 768, 3072), ("tall  M=200704 N=768 K=768 ", 200704, 768, 768)]:
    dy = torch.randn(M, N, device=dev).to(bf)
    x = torch.randn(M, Kd, device=dev).to(bf)
    ms = timeit(lambda: K.linear_bwd_weight(dy, x, want_bias=False), 20)
    print(f"{name}: {ms * 1000:8.1f} us  (debug={os.environ.get('TAD_GEMM_DEBUG', '0')})", flush=True)
