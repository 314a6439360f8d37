#!/usr/bin/env python3
"""Interleaved A/B of the weight-gradient entry point (tad_linear_bwd_weight) of several builds of the library in ONE process, at the four
dW shapes of a ViT-B block (M = 50176):   python tools/ab_tn.py name=path ... [--rounds 7] [--iters 10]"""
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


def bind(spec):
    name, path = spec.split("=", 1)
    lib = C.CDLL(path)
    for fn in ("tad_linear_bwd_weight", "tad_linear_bwd_weight_workspace_bytes"):
        rt, at = _lib.SIGNATURES[fn]
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = rt, at
    return name, lib


libs = [bind(s) for s in a.libs]
shapes = (("qkv  [3D, D]", 3 * D, D), ("proj [D, D]", D, D), ("fc1  [4D, D]", 4 * D, D), ("fc2  [D, 4D]", D, 4 * D))
wsb = max(int(lib.tad_linear_bwd_weight_workspace_bytes(M, n, k)) for _, lib in libs for _, n, k in shapes)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        rc = fn()
    e.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    return s.elapsed_time(e) / a.iters * 1e3


tot = {n: 0.0 for n, _ in libs}
print(f"{'dW shape':16s}" + "".join(f"{n + ' us':>12s}{'TF':>7s}" for n, _ in libs) + "   bit-identical to the first")
for label, n, k in shapes:
    dy, x = torch.randn(M, n, device=dev).to(bf), torch.randn(M, k, device=dev).to(bf)
    outs = []
    fns = {}
    for name, lib in libs:
        dW, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
        fns[name] = (lambda lib=lib, dW=dW, db=db: lib.tad_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), dW.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), wsb, M, n, k, st))
        for _ in range(3):
            assert fns[name]() == 0
        torch.cuda.synchronize()
        outs.append((dW.clone(), db.clone()))
    same = all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs)
    t = {name: [] for name, _ in libs}
    for _ in range(a.rounds):
        for name, _ in libs:
            t[name].append(timeit(fns[name]))
    line = f"{label:16s}"
    for name, _ in libs:
        med = statistics.median(t[name])
        tot[name] += med
        line += f"{med:12.1f}{2.0 * M * n * k / med / 1e6:7.0f}"
    print(line + f"   {same}", flush=True)
print(f"{'sum':16s}" + "".join(f"{tot[n]:12.1f}{'':7s}" for n, _ in libs))
