#!/usr/bin/env python3
"""The planned launch of the bias-only Linears of a ViT-B block with its whole rounds on the eight-wave (w4_plain = 0) or the four-wave 256 x 256
kernel (tad_linear_tuning("w4_plain", K_min)), interleaved in one process; results must be bit-identical.  python tools/exp_w4_plain.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402
M, D, dev, bf = 50176, 768, "cuda", torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)  # noqa: E731
x_d, x_3d, x_4d = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
W = {n: rnd(*s, scale=0.02) for n, s in {"qkv": (3 * D, D), "proj": (D, D), "fc1T": (D, 4 * D), "qkvT": (D, 3 * D)}.items()}
b_d = torch.randn(D, device=dev)
cases = [("qkv fwd (q prescale) K768", lambda: K.linear_fwd_qkv(x_d, W["qkv"], b_d, b_d, q_prescale=0.18)),
         ("dX proj 16-bit K768", lambda: K.linear_bwd_input(x_d, W["proj"])),
         ("dX qkv 16-bit K2304", lambda: K.linear_bwd_input(x_3d, W["qkvT"])),
         ("dX fc1 16-bit K3072", lambda: K.linear_bwd_input(x_4d, W["fc1T"]))]


def timeit(fn, it=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


print(f"{'case':28s} {'8-wave us':>10s} {'4-wave us':>10s}  bit-identical")
try:
    for label, fn in cases:
        outs, t = [], [[], []]
        for v in (0, 128):
            K.linear_tuning(w4_plain=v)
            outs.append(fn().clone())
            fn(); fn()
        for _ in range(7):
            for i, v in enumerate((0, 128)):
                K.linear_tuning(w4_plain=v)
                t[i].append(timeit(fn))
        print(f"{label:28s} {statistics.median(t[0]):10.1f} {statistics.median(t[1]):10.1f}  {torch.equal(outs[0], outs[1])}", flush=True)
finally:
    K.linear_tuning(w4_plain=K.LINEAR_TUNING_DEFAULTS["w4_plain"])
