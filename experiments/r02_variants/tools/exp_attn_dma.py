#!/usr/bin/env python3
"""A/B of the attention kernels' LDS-DMA placement (tad_attn_tuning "dma_mode") at the benchmark shape (B = 32, N = 1568, H = 12),
random data, interleaved rounds in one process; checks that both placements give bit-identical results.
    python tools/exp_attn_dma.py [--modes 0,1] [--rounds 7]          (mode 2 = no in-loop DMA, timing only, ablation builds)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="0,1")
ap.add_argument("--configs", default="", help="';'-separated 'key=value,key=value' tuning sets (tad_attn_tuning), e.g. 'bwd_stages=2;bwd_stages=3'")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--B", type=int, default=32)
a = ap.parse_args()
DEFAULTS = dict(dma_mode=0, bwd_stages=2, dkv_keys=32)
if a.configs:
    modes = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",")) for c in a.configs.split(";")]
else:
    modes = [dict(dma_mode=int(m)) for m in a.modes.split(",")]


def apply(cfg):
    K.attn_tuning(**{**DEFAULTS, **cfg})
B, N, H = a.B, 1568, 12
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
apply({})
out0, lse = K.attn_fwd(qkv, B, N, H, 0.125)
dout = torch.randn_like(out0)
dq0 = K.attn_bwd(qkv, out0, dout, lse, B, N, H, 0.125)


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


fns = {"attn_fwd": lambda: K.attn_fwd(qkv, B, N, H, 0.125), "attn_bwd": lambda: K.attn_bwd(qkv, out0, dout, lse, B, N, H, 0.125)}
t = {k: np.zeros((len(modes), a.rounds)) for k in fns}
for m in modes:
    if m.get("dma_mode") in (2, 3):
        continue
    apply(m)
    o, _ = K.attn_fwd(qkv, B, N, H, 0.125)
    d = K.attn_bwd(qkv, out0, dout, lse, B, N, H, 0.125)
    print(f"mode {m}: forward bit-identical to mode 0: {torch.equal(o, out0)}, backward: {torch.equal(d, dq0)}", flush=True)
for r in range(a.rounds):
    for i, m in enumerate(modes):
        apply(m)
        for k, fn in fns.items():
            fn()
            t[k][i, r] = timed(fn)
apply({})
for k in fns:
    fl = (4.0 if k == "attn_fwd" else 8.0) * B * H * N * N * 64
    print(k, " | ".join(f"mode {m}: median {np.median(t[k][i]):7.1f} us (min {t[k][i].min():7.1f}) {fl / np.median(t[k][i]) / 1e6:6.0f} TF" for i, m in enumerate(modes)))
