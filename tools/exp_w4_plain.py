#!/usr/bin/env python3
"""The planned launch of the bias-only / residual Linears of a ViT block with its whole rounds on the eight-wave (w4_plain = 0, w4_epilogues = 0) or the
four-wave 256 x 256 kernel (tad_linear_tuning("w4_plain", K_min) / ("w4_epilogues", 4)), interleaved in one process; results must be bit-identical.
    python tools/exp_w4_plain.py [--D 768] [--rows 50176]      (D = 384: ViT-S, 512: the MAE decoder, 1024: ViT-L)"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--rows", type=int, default=50176)
a = ap.parse_args()
M, D, dev, bf = a.rows, a.D, "cuda", torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)  # noqa: E731
x_d, x_3d, x_4d = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
W = {n: rnd(*s, scale=0.02) for n, s in {"qkv": (3 * D, D), "proj": (D, D), "fc1T": (D, 4 * D), "qkvT": (D, 3 * D)}.items()}
b_d, res = torch.randn(D, device=dev), torch.randn(M, D, device=dev)
first = lambda r: r[0] if isinstance(r, tuple) else r  # noqa: E731
cases = [(f"qkv fwd (q prescale) K{D}", lambda: K.linear_fwd_qkv(x_d, W["qkv"], b_d, b_d, q_prescale=0.18)),
         (f"dX proj 16-bit K{D}", lambda: K.linear_bwd_input(x_d, W["proj"])),
         (f"dX qkv f32 K{3 * D}", lambda: K.linear_bwd_input(x_3d, W["qkvT"], out_dtype=torch.float32)),
         (f"dX fc1 f32 K{4 * D}", lambda: K.linear_bwd_input(x_4d, W["fc1T"], out_dtype=torch.float32)),
         (f"proj fwd +res f32 K{D}", lambda: K.linear_fwd(x_d, W["proj"], b_d, out_dtype=torch.float32, epilogue=2, residual=res)),
         (f"fc2 fwd +res f32 K{4 * D}", lambda: K.linear_fwd(x_4d, W["fc1T"], b_d, out_dtype=torch.float32, epilogue=2, residual=res))]
CFG = [dict(w4_plain=0, w4_epilogues=0), dict(w4_plain=128, w4_epilogues=4)]


def timeit(fn, it=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


print(f"D {D} rows {M}\n{'case':28s} {'8-wave us':>10s} {'4-wave us':>10s}  bit-identical")
try:
    for label, fn in cases:
        outs, t = [], [[], []]
        for cfg in CFG:
            K.linear_tuning(**cfg)
            outs.append(first(fn()).clone())
            fn(); fn()
        for _ in range(7):
            for i, cfg in enumerate(CFG):
                K.linear_tuning(**cfg)
                t[i].append(timeit(fn))
        print(f"{label:28s} {statistics.median(t[0]):10.1f} {statistics.median(t[1]):10.1f}  {torch.equal(outs[0], outs[1])}", flush=True)
finally:
    K.linear_tuning(**{k: K.LINEAR_TUNING_DEFAULTS[k] for k in ("w4_plain", "w4_epilogues")})
