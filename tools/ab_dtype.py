#!/usr/bin/env python3
"""bf16 kernels against their IEEE-half twins, interleaved in ONE process on ONE device (cdna_hip_programming.md rule 24), at the in-model
shapes of a ViT-B block (M = 50176): where does the `half` mode lose its 5-7 % against `fast`?

    python tools/ab_dtype.py [--rounds 7] [--iters 10] [--only gemm_nt,gemm_tn,attn,ln]

Same values in both formats (N(0,1) activations, N(0, 0.02) weights: all exactly representable ranges), random data, HIP events.
"""
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
ap.add_argument("--only", default="")
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--grad-scale", type=float, default=1.0, help="multiply the gradient-side operands (dy, dout) by this factor: the half mode's "
                "gradients carry the loss scale (65536), the bf16 mode's do not -- does the operand magnitude change the time?")
a = ap.parse_args()
only = set(x for x in a.only.split(",") if x)
dev = "cuda"
B, N, D = a.B, 1568, a.D
H = D // 64
M = B * N
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def want(name):
    return not only or name in only


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3  # us


cases = []  # (label, flops, {dtype name: fn})


def both(label, flops, make):
    cases.append((label, flops, {k: make(v) for k, v in DT.items()}))


g = torch.Generator(device=dev)
g.manual_seed(0)


def rnd32(*shape, scale=1.0):
    return torch.randn(*shape, device=dev, generator=g) * scale


if want("gemm_nt"):
    x_d32, x_4d32 = rnd32(M, D), rnd32(M, 4 * D)
    dy_d32, dy_3d32, dy_4d32 = rnd32(M, D) * a.grad_scale, rnd32(M, 3 * D) * a.grad_scale, rnd32(M, 4 * D) * a.grad_scale
    w32 = {n: rnd32(*s, scale=0.02) for n, s in {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D),
                                                 "fc2T": (4 * D, D), "fc1T": (D, 4 * D), "qkvT": (D, 3 * D)}.items()}
    b_d, b_4d = torch.randn(D, device=dev), torch.randn(4 * D, device=dev)
    res = torch.randn(M, D, device=dev)
    hpre32 = rnd32(M, 4 * D)

    def mk_qkv(dt):
        x, w = x_d32.to(dt), w32["qkv"].to(dt)
        return lambda: K.linear_fwd_qkv(x, w, b_d, b_d, q_prescale=0.18)

    def mk_res(xn, wn):
        def mk(dt):
            x, w = (x_d32 if xn == "d" else x_4d32).to(dt), w32[wn].to(dt)
            return lambda: K.linear_fwd(x, w, b_d, out_dtype=torch.float32, epilogue=2, residual=res)
        return mk

    def mk_fc1(dt):
        x, w = x_d32.to(dt), w32["fc1"].to(dt)
        return lambda: K.linear_fwd(x, w, b_4d, epilogue=1, want_preact=True)

    def mk_dx(dyt, wn, out32, pre=False):
        def mk(dt):
            dy, w = dyt.to(dt), w32[wn].to(dt)
            hp = hpre32.to(dt) if pre else None
            return lambda: K.linear_bwd_input(dy, w, out_dtype=torch.float32 if out32 else None, gelu_preact=hp)
        return mk

    both("qkv fwd (q prescale)", 2.0 * M * 3 * D * D, mk_qkv)
    both("proj fwd +res f32", 2.0 * M * D * D, mk_res("d", "proj"))
    both("fc1 fwd gelu +preact", 2.0 * M * 4 * D * D, mk_fc1)
    both("fc2 fwd +res f32", 2.0 * M * 4 * D * D, mk_res("4d", "fc2"))
    both("dX fc2 (dgelu) 16-bit", 2.0 * M * 4 * D * D, mk_dx(dy_d32, "fc2T", False, pre=True))
    both("dX fc1 f32", 2.0 * M * 4 * D * D, mk_dx(dy_4d32, "fc1T", True))
    both("dX proj 16-bit", 2.0 * M * D * D, mk_dx(dy_d32, "proj", False))
    both("dX qkv f32", 2.0 * M * 3 * D * D, mk_dx(dy_3d32, "qkvT", True))

if want("gemm_tn"):
    for name, n, k in (("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", 4 * D, D), ("dW fc2", D, 4 * D)):
        dy32, x32 = rnd32(M, n) * a.grad_scale, rnd32(M, k)

        def mk(dt, dy32=dy32, x32=x32):
            dy, x = dy32.to(dt), x32.to(dt)
            return lambda: K.linear_bwd_weight(dy, x, want_bias=True)
        both(name, 2.0 * M * n * k, mk)
        del dy32, x32

if want("attn"):
    qkv32 = rnd32(M, 3 * D)
    qkv32[:, :D] *= K.q_prescale_of(0.125)
    dout32 = rnd32(M, D) * a.grad_scale

    def mk_f(dt):
        t = qkv32.to(dt)
        return lambda: K.attn_fwd(t, B, N, H, 0.125, want_lo=True, q_prescaled=True)

    def mk_b(dt):
        t, do = qkv32.to(dt), dout32.to(dt)
        out, lse, lo = K.attn_fwd(t, B, N, H, 0.125, want_lo=True, q_prescaled=True)
        return lambda: K.attn_bwd(t, out, do, lse, B, N, H, 0.125, out_lo=lo, q_prescaled=True)
    both("attn fwd (+out_lo)", 4.0 * B * H * N * N * 64, mk_f)
    both("attn bwd (dq + dkv)", 8.0 * B * H * N * N * 64, mk_b)

if want("ln"):
    x = torch.randn(M, D, device=dev)
    gm, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    _, mean, rstd = K.layernorm_fwd(x, gm, bt, 1e-6)
    dy32 = rnd32(M, D) * a.grad_scale

    def mk_lf(dt):
        return lambda: K.layernorm_fwd(x, gm, bt, 1e-6, out_dtype=dt)

    def mk_lb(dt):
        dy = dy32.to(dt)
        return lambda: K.layernorm_bwd(dy, x, gm, mean, rstd, dres=x, want_bf16=True, want_colsum=True)
    both("layernorm fwd", 0.0, mk_lf)
    both("layernorm bwd (16-bit dy)", 0.0, mk_lb)

tot = {k: 0.0 for k in DT}
print(f"B={B} D={D} M={M} grad_scale={a.grad_scale:g} rounds={a.rounds} iters={a.iters}  (median us over interleaved rounds)")
print(f"{'case':28s}" + "".join(f"{n + ' us':>11s}{'TF/s':>7s}" for n in DT) + f"{'f16/bf16':>10s}")
for label, flops, fns in cases:
    for fn in fns.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    t = {k: [] for k in DT}
    for _ in range(a.rounds):
        for k in DT:
            t[k].append(timeit(fns[k]))
    med = {k: statistics.median(t[k]) for k in DT}
    for k in DT:
        tot[k] += med[k]
    print(f"{label:28s}" + "".join(f"{med[k]:11.1f}{(flops / med[k] / 1e6 if flops else 0):7.0f}" for k in DT) + f"{med['f16'] / med['bf16']:10.3f}", flush=True)
print(f"{'sum':28s}" + "".join(f"{tot[k]:11.1f}{'':7s}" for k in DT) + f"{tot['f16'] / tot['bf16']:10.3f}")
