#!/usr/bin/env python3
"""Interleaved A/B of the Linear entry points of several builds of the library in ONE process (cdna_hip_programming.md rule 24), at the
in-model shapes and epilogues of a ViT-B block (M = 50176):   python tools/ab_gemm.py name=path ... [--rounds 7] [--iters 10]
Each library is loaded by path with ctypes; signatures from simple_tad_amd/_lib.py (ABI 4)."""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--M", type=int, default=50176)
a = ap.parse_args()
M, D, dev, bf = a.M, a.D, "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream


def rnd(*shape, scale=1.0, dtype=bf):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def bind(spec):
    name, path = spec.split("=", 1)
    lib = C.CDLL(path)
    for fn in ("tad_linear_fwd", "tad_linear_bwd_input", "tad_linear_fwd_qkv", "tad_linear_workspace_bytes"):
        rt, at = _lib.SIGNATURES[fn]
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = rt, at
    return name, lib


libs = [bind(s) for s in a.libs]
wsb = max(int(lib.tad_linear_workspace_bytes(M, n_, k_)) for _, lib in libs for n_, k_ in ((D, D), (D, 3 * D), (D, 4 * D), (4 * D, D), (3 * D, D)))
ws = torch.zeros(max(wsb, 8), dtype=torch.uint8, device=dev)
cases = []
# (label, flops, fn(lib))
x_d, x_4d = rnd(M, D), rnd(M, 4 * D)
w_qkv, w_proj, w_fc1, w_fc2 = rnd(3 * D, D, scale=.02), rnd(D, D, scale=.02), rnd(4 * D, D, scale=.02), rnd(D, 4 * D, scale=.02)
wT_fc2, wT_fc1, wT_qkv = rnd(4 * D, D, scale=.02), rnd(D, 4 * D, scale=.02), rnd(D, 3 * D, scale=.02)  # as [K_out, N_in] operands of dX
b_d, b_3d, b_4d = torch.randn(D, device=dev), torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev)
res = torch.randn(M, D, device=dev)
y_d32, y_3d, y_4d, h_4d, y_d16 = torch.empty(M, D, device=dev), torch.empty(M, 3 * D, device=dev, dtype=bf), torch.empty(M, 4 * D, device=dev, dtype=bf), torch.empty(M, 4 * D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)
dy_3d = rnd(M, 3 * D)
hpre = rnd(M, 4 * D)
p = lambda t: t.data_ptr()  # noqa: E731
cases.append(("qkv fwd (q prescale)", 2.0 * M * 3 * D * D, lambda L: L.tad_linear_fwd_qkv(p(x_d), p(w_qkv), p(b_d), p(b_d), p(y_3d), 1, 0.18, M, 3 * D, D, st)))
cases.append(("proj fwd +res f32", 2.0 * M * D * D, lambda L: L.tad_linear_fwd(p(x_d), p(w_proj), p(b_d), p(y_d32), 0, 2, None, p(res), None, None, 1, p(ws), wsb, M, D, D, st)))
cases.append(("fc1 fwd gelu +preact", 2.0 * M * 4 * D * D, lambda L: L.tad_linear_fwd(p(x_d), p(w_fc1), p(b_4d), p(y_4d), 1, 1, p(h_4d), None, None, None, 1, p(ws), wsb, M, 4 * D, D, st)))
cases.append(("fc2 fwd +res f32", 2.0 * M * 4 * D * D, lambda L: L.tad_linear_fwd(p(x_4d), p(w_fc2), p(b_d), p(y_d32), 0, 2, None, p(res), None, None, 1, p(ws), wsb, M, D, 4 * D, st)))
cases.append(("dX fc2 (dgelu) bf16", 2.0 * M * 4 * D * D, lambda L: L.tad_linear_bwd_input(p(x_d), p(wT_fc2), p(y_4d), 1, p(hpre), p(ws), wsb, M, D, 4 * D, st)))
cases.append(("dX fc1 f32", 2.0 * M * 4 * D * D, lambda L: L.tad_linear_bwd_input(p(x_4d), p(wT_fc1), p(y_d32), 0, None, p(ws), wsb, M, 4 * D, D, st)))
cases.append(("dX proj bf16", 2.0 * M * D * D, lambda L: L.tad_linear_bwd_input(p(x_d), p(w_proj), p(y_d16), 1, None, p(ws), wsb, M, D, D, st)))
cases.append(("dX qkv f32", 2.0 * M * 3 * D * D, lambda L: L.tad_linear_bwd_input(p(dy_3d), p(wT_qkv), p(y_d32), 0, None, p(ws), wsb, M, 3 * D, D, st)))


def timeit(fn, lib):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        rc = fn(lib)
    e.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    return s.elapsed_time(e) / a.iters * 1e3


tot = {n: 0.0 for n, _ in libs}
print(f"{'case':26s}" + "".join(f"{n + ' us':>12s}{'TF':>7s}" for n, _ in libs))
for label, flops, fn in cases:
    for _, lib in libs:
        for _ in range(3):
            fn(lib)
    t = {n: [] for n, _ in libs}
    for _ in range(a.rounds):
        for n, lib in libs:
            t[n].append(timeit(fn, lib))
    line = f"{label:26s}"
    for n, _ in libs:
        med = statistics.median(t[n])
        tot[n] += med
        line += f"{med:12.1f}{flops / med / 1e6:7.0f}"
    print(line, flush=True)
print(f"{'sum':26s}" + "".join(f"{tot[n]:12.1f}{'':7s}" for n, _ in libs))
