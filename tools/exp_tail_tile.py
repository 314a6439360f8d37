#!/usr/bin/env python3
"""The tail of the split plan (the rows behind the whole rounds of 256 x 256 tiles: 6656 of ViT-B's 50176 at N = 768) as 256 x 128 tiles (variant 3:
156 workgroups on 256 CUs), 192 x 128 (variant 8: 210) or 128 x 128 (variant 2: 312 on two slots per CU), interleaved in one process; outputs must
be bit-identical.  python tools/exp_tail_tile.py [--rows 6656]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=6656)
ap.add_argument("--variants", default="3,8,2")
a = ap.parse_args()
M, D, dev, bf = a.rows, 768, "cuda", torch.bfloat16
VAR = [int(v) for v in a.variants.split(",")]
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(bf)  # noqa: E731
x_d, x_3d, x_4d = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
W = {n: rnd(*s, scale=0.02) for n, s in {"proj": (D, D), "fc2": (D, 4 * D), "qkvT": (D, 3 * D), "fc1": (4 * D, D), "qkv": (3 * D, D)}.items()}
b_d, b_4d, res = torch.randn(D, device=dev), torch.randn(4 * D, device=dev), torch.randn(M, D, device=dev)
cases = [("proj fwd +res f32 K768", lambda: K.linear_fwd(x_d, W["proj"], b_d, out_dtype=torch.float32, epilogue=2, residual=res)),
         ("fc2 fwd +res f32 K3072", lambda: K.linear_fwd(x_4d, W["fc2"], b_d, out_dtype=torch.float32, epilogue=2, residual=res)),
         ("dX proj 16-bit K768", lambda: K.linear_bwd_input(x_d, W["proj"])),
         ("dX qkv f32 K2304", lambda: K.linear_bwd_input(x_3d, W["qkvT"], out_dtype=torch.float32)),
         ("dX fc1 f32 K3072", lambda: K.linear_bwd_input(x_4d, W["fc2"], out_dtype=torch.float32)),
         ("fc1 fwd gelu +preact K768", lambda: K.linear_fwd(x_d, W["fc1"], b_4d, epilogue=1, want_preact=True)),
         ("dX fc2 dgelu K768", lambda: K.linear_bwd_input(x_d, W["fc1"], gelu_preact=x_4d)),
         ("qkv fwd K768", lambda: K.linear_fwd_qkv(x_d, W["qkv"], b_d, b_d, q_prescale=0.18))]


def timeit(fn, it=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


print(f"rows {M}\n{'case':28s}" + "".join(f"{'v' + str(v) + ' us':>10s}" for v in VAR) + "  bit-identical")
tot = [0.0] * len(VAR)
try:
    for label, fn in cases:
        outs, t = [], [[] for _ in VAR]
        for v in VAR:
            K.linear_tuning(variant=v)
            r = fn(); outs.append((r[0] if isinstance(r, tuple) else r).clone())
            fn(); fn()
        for _ in range(7):
            for i, v in enumerate(VAR):
                K.linear_tuning(variant=v)
                t[i].append(timeit(fn))
        med = [statistics.median(x) for x in t]
        tot = [p + q for p, q in zip(tot, med)]
        print(f"{label:28s}" + "".join(f"{m:10.1f}" for m in med) + f"  {all(torch.equal(outs[0], o) for o in outs[1:])}", flush=True)
    print(f"{'sum':28s}" + "".join(f"{m:10.1f}" for m in tot))
finally:
    K.linear_tuning(variant=0)
