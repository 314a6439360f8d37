#!/usr/bin/env python3
"""Can the under-filled tail launch of an N = 768 Linear (588 tiles of 256 x 256 = 2.3 rounds of one workgroup per CU: 2 full rounds +
156 workgroups of 256 x 128) run BESIDE the HBM-bound LayerNorm that follows it, instead of in front of it?  Rows are independent in
both kernels, so the LayerNorm of the main rows only needs the main launch: [main GEMM] -> {tail GEMM on a side stream || LayerNorm of
the main rows} -> LayerNorm of the tail rows.  Times the four Linear -> LayerNorm pairs of a ViT-B block both ways (interleaved rounds,
one process, random data).    python tools/exp_tail_overlap.py [--rounds 7] [--iters 10]"""
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
ap.add_argument("--main-panels", type=int, default=170)
a = ap.parse_args()
dev = "cuda"
B, N, D = 32, 1568, 768
M = B * N
MAIN = a.main_panels * 256
bf = torch.bfloat16
torch.manual_seed(0)
side = torch.cuda.Stream()


def rnd(*s, scale=1.0, dtype=bf):
    return (torch.randn(*s, device=dev) * scale).to(dtype)


def pair_fwd(Kd):
    x, w, b = rnd(M, Kd), rnd(D, Kd, scale=0.02), torch.randn(D, device=dev)
    res = torch.randn(M, D, device=dev)
    g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)

    def seq():
        y, _ = K.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=res)
        return K.layernorm_fwd(y, g, be, 1e-6)

    def ovl():
        main = torch.cuda.current_stream()
        ym, _ = K.linear_fwd(x[:MAIN], w, b, out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=res[:MAIN])
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            yt, _ = K.linear_fwd(x[MAIN:], w, b, out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=res[MAIN:])
            ev2 = torch.cuda.Event()
            ev2.record(side)
        o1 = K.layernorm_fwd(ym, g, be, 1e-6)
        main.wait_event(ev2)
        o2 = K.layernorm_fwd(yt, g, be, 1e-6)
        return o1, o2
    return seq, ovl


def pair_bwd(Kd):
    dy, wT = rnd(M, Kd), rnd(D, Kd, scale=0.02)   # dX [M, D] = dy [M, Kd] @ W  (wT given as [D, Kd])
    x = torch.randn(M, D, device=dev)
    gres = torch.randn(M, D, device=dev)
    g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    _, mean, rstd = K.layernorm_fwd(x, g, be, 1e-6)

    def seq():
        dxn = K.linear_bwd_input(dy, wT, out_dtype=torch.float32)
        return K.layernorm_bwd(dxn, x, g, mean, rstd, dres=gres, want_bf16=True, want_colsum=True)

    def ovl():
        main = torch.cuda.current_stream()
        dm = K.linear_bwd_input(dy[:MAIN], wT, out_dtype=torch.float32)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            dt_ = K.linear_bwd_input(dy[MAIN:], wT, out_dtype=torch.float32)
            ev2 = torch.cuda.Event()
            ev2.record(side)
        r1 = K.layernorm_bwd(dm, x[:MAIN], g, mean[:MAIN], rstd[:MAIN], dres=gres[:MAIN], want_bf16=True, want_colsum=True)
        main.wait_event(ev2)
        r2 = K.layernorm_bwd(dt_, x[MAIN:], g, mean[MAIN:], rstd[MAIN:], dres=gres[MAIN:], want_bf16=True, want_colsum=True, into=(r1[2], r1[3], r1[4]))
        return r1, r2
    return seq, ovl


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


print(f"{'pair':44s} {'sequential us':>14s} {'tail beside LN us':>18s} {'saved':>8s}")
tot = [0.0, 0.0]
for name, mk, Kd in (("proj fwd (K 768)  -> LayerNorm fwd", pair_fwd, 768), ("fc2 fwd (K 3072) -> LayerNorm fwd", pair_fwd, 3072),
                     ("dX(fc1) (K 3072)  -> LayerNorm bwd", pair_bwd, 3072), ("dX(qkv) (K 2304)  -> LayerNorm bwd", pair_bwd, 2304)):
    seq, ovl = mk(Kd)
    for _ in range(3):
        seq(), ovl()
    torch.cuda.synchronize()
    ts, to = [], []
    for _ in range(a.rounds):
        ts.append(timeit(seq))
        to.append(timeit(ovl))
    ms, mo = statistics.median(ts), statistics.median(to)
    tot[0] += ms
    tot[1] += mo
    print(f"{name:44s} {ms:14.1f} {mo:18.1f} {ms - mo:8.1f}")
    del seq, ovl
    torch.cuda.empty_cache()
print(f"{'sum':44s} {tot[0]:14.1f} {tot[1]:18.1f} {tot[0] - tot[1]:8.1f}   (x 12 blocks = {(tot[0] - tot[1]) * 12e-3:.2f} ms per step)")
