#!/usr/bin/env python3
"""Calibration, not a product path: torch's own fused attention (torch.nn.functional.scaled_dot_product_attention -> the ROCm flash /
memory-efficient backends that ship with this torch build) at the benchmark's attention shape (B 32, H 12, N 1568, d 64, bf16, random data)
beside this library's forward and backward kernels; interleaved rounds in one process.

    python tools/exp_sdpa_compare.py [--rounds 7] [--iters 5]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
B, H, N, d = 32, 12, 1568, 64
scale = d ** -0.5


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


qkv = torch.randn(B * N, 3 * H * d, device=dev).to(bf)
dout = torch.randn(B * N, H * d, device=dev).to(bf)
out, lse = K.attn_fwd(qkv, B, N, H, scale)
# the vendor route gets its preferred layout for free: [B, H, N, d] contiguous q, k, v (no permute inside the timed region)
q5 = qkv.view(B, N, 3, H, d).permute(2, 0, 3, 1, 4).contiguous()
q, k, v = (t.clone().requires_grad_() for t in q5)
do_bhnd = dout.view(B, N, H, d).permute(0, 2, 1, 3).contiguous()
ref = F.scaled_dot_product_attention(q, k, v)
err = ((ref.detach().permute(0, 2, 1, 3).reshape(B * N, H * d).float() - out.float()).norm() / out.float().norm()).item()
print(f"forward outputs agree to rel-L2 {err:.2e}")


def vendor_fwd():
    with torch.no_grad():
        F.scaled_dot_product_attention(q, k, v)


def vendor_fwd_bwd():
    o = F.scaled_dot_product_attention(q, k, v)
    o.backward(do_bhnd)
    q.grad = k.grad = v.grad = None


fns = {"vendor fwd": vendor_fwd, "tad attn_fwd": lambda: K.attn_fwd(qkv, B, N, H, scale), "vendor fwd+bwd": vendor_fwd_bwd,
       "tad attn_bwd": lambda: K.attn_bwd(qkv, out, dout, lse, B, N, H, scale)}
for f in fns.values():
    for _ in range(2):
        f()
t = {kk: [] for kk in fns}
for _ in range(a.rounds):
    for kk, f in fns.items():
        t[kk].append(timed(f, a.iters))
med = {kk: float(np.median(vv)) for kk, vv in t.items()}
fl_f, fl_b = 4.0 * B * H * N * N * d, 8.0 * B * H * N * N * d
print(f"vendor fwd      {med['vendor fwd']:8.1f} us  {fl_f / med['vendor fwd'] / 1e6:6.0f} TF")
print(f"tad attn_fwd    {med['tad attn_fwd']:8.1f} us  {fl_f / med['tad attn_fwd'] / 1e6:6.0f} TF")
vb = med["vendor fwd+bwd"] - med["vendor fwd"]
print(f"vendor bwd      {vb:8.1f} us  {fl_b / vb / 1e6:6.0f} TF   (fwd+bwd {med['vendor fwd+bwd']:.1f} us minus fwd)")
print(f"tad attn_bwd    {med['tad attn_bwd']:8.1f} us  {fl_b / med['tad attn_bwd'] / 1e6:6.0f} TF")
print("torch", torch.__version__)
