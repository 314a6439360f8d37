#!/usr/bin/env python3
"""A/B of the Linear GEMM scheduling knobs (tad_linear_tuning) at the real ViT-B/16 shapes (M = 50176 rows), random data, interleaved
rounds in ONE process (cdna_hip_programming.md rule 24): for every shape, every configuration is timed `rounds` times in rotation and the
median / min per configuration is reported.

    python tools/exp_gemm_knobs.py --knob group_m --values 0,2,4,8,16,32,64 [--rounds 7] [--iters 10]
    python tools/exp_gemm_knobs.py --configs "split_tail=1;split_tail=0"
Under rocprofv3 (--pmc FETCH_SIZE --kernel-trace) use --trace: every (shape, config) is launched exactly 3 times in the printed order.
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
ap.add_argument("--knob", default="group_m")
ap.add_argument("--values", default="0,2,4,8,16,32,64")
ap.add_argument("--configs", default="", help="';'-separated 'k=v,k=v' configurations (overrides --knob/--values)")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--shapes", default="")
ap.add_argument("--trace", action="store_true")
ap.add_argument("--D", type=int, default=768, help="model width (384: ViT-S, 512: MAE decoder, 1024: ViT-L)")
ap.add_argument("--check", action="store_true", help="every configuration's outputs must be bit-identical to the first one's")
a = ap.parse_args()
cfgs = ([dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",")) for c in a.configs.split(";")] if a.configs
        else [{a.knob: int(v)} for v in a.values.split(",")])
dev, bf, D, M = "cuda", torch.bfloat16, a.D, 50176
SHAPES = [(f"qkv fwd   N{3 * D} K{D} plain", 3 * D, D, "plain"), (f"proj fwd  N{D} K{D} res", D, D, "res"),
          (f"fc1 fwd   N{4 * D} K{D} gelu", 4 * D, D, "gelu"), (f"fc2 fwd   N{D} K{4 * D} res", D, 4 * D, "res"),
          (f"dX qkv    N{D} K{3 * D} plain", D, 3 * D, "plain"), (f"dX fc2    N{4 * D} K{D} dgelu", 4 * D, D, "dgelu"),
          (f"dX fc1    N{D} K{4 * D} plain", D, 4 * D, "plain"), (f"dX proj   N{D} K{D} plain", D, D, "plain")]
if a.shapes:
    SHAPES = [s for s in SHAPES if any(s[0].startswith(p) for p in a.shapes.split(","))]


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


total = {i: 0.0 for i in range(len(cfgs))}
for name, n, k, mode in SHAPES:
    x = torch.randn(M + M // 8 + 8, k, device=dev).to(bf)[:M]  # (over-allocated: the ablation build's debug bit 128 reads rows 128 bytes further apart)
    w = (torch.randn(n, k, device=dev) * 0.02).to(bf)
    bias = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev) if mode == "res" else None
    h = torch.randn(M, n, device=dev).to(bf) if mode == "dgelu" else None
    if mode == "dgelu":
        fn = lambda: K.linear_bwd_input(x, w, gelu_preact=h)  # noqa: E731
    elif mode == "gelu":
        fn = lambda: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)  # noqa: E731
    elif mode == "res":
        fn = lambda: K.linear_fwd(x, w, bias, out_dtype=torch.float32, epilogue=2, residual=res)  # noqa: E731
    else:
        fn = lambda: K.linear_fwd(x, w, bias)  # noqa: E731
    if a.trace:
        for c in cfgs:
            K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, **c})
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            print(f"TRACE {name} | {c}", flush=True)
        continue
    for _ in range(3):
        fn()
    if a.check:
        outs = []
        for c in cfgs:
            K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, **c})
            o = fn()
            outs.append([t_.clone() for t_ in (o if isinstance(o, tuple) else (o,)) if t_ is not None])
        for i, o in enumerate(outs[1:], 1):
            for u, v in zip(outs[0], o):
                assert torch.equal(u, v), f"{name}: configuration {cfgs[i]} differs from {cfgs[0]}: max |d| {(u.float() - v.float()).abs().max().item():.3e}"
        del outs
    t = np.zeros((len(cfgs), a.rounds))
    for r in range(a.rounds):
        for i, c in enumerate(cfgs):
            K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, **c})
            fn()
            t[i, r] = timed(fn, a.iters)
    gf = 2.0 * M * n * k / 1e9
    line = f"{name:30s}"
    for i, c in enumerate(cfgs):
        med = float(np.median(t[i]))
        total[i] += med
        line += f" | {','.join(f'{kk}={vv}' for kk, vv in c.items())}: {med:7.1f} us (min {t[i].min():7.1f}) {gf / med * 1e3:6.0f} TF"
    print(line, flush=True)
    del x, w, res, h
K.linear_tuning(**K.LINEAR_TUNING_DEFAULTS)
if not a.trace:
    print("sum of medians per configuration (us):", "  ".join(f"{cfgs[i]}: {v:.1f}" for i, v in total.items()))
