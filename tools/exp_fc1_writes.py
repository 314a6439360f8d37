#!/usr/bin/env python3
"""Is the GELU Linear (fc1 forward: M x 3072 x 768, two 16-bit outputs = 616 MB) bound by its output stores?  The same launch with and without
the pre-activation copy, and the bias-only launch of the same shape (one output, no GELU), interleaved.  python tools/exp_fc1_writes.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_tad_amd import kernels as K  # noqa: E402
M, D, dev, bf = 50176, 768, "cuda", torch.bfloat16
x = torch.randn(M, D, device=dev).to(bf)
w = (torch.randn(4 * D, D, device=dev) * 0.02).to(bf)
b = torch.randn(4 * D, device=dev)
cases = [("gelu + pre-activation (616 MB out)", lambda: K.linear_fwd(x, w, b, epilogue=1, want_preact=True)),
         ("gelu only (308 MB out)", lambda: K.linear_fwd(x, w, b, epilogue=1, want_preact=False)),
         ("bias only (308 MB out)", lambda: K.linear_fwd(x, w, b))]


def timeit(fn, it=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for _, fn in cases:
    fn(); fn()
t = [[] for _ in cases]
for _ in range(7):
    for i, (_, fn) in enumerate(cases):
        t[i].append(timeit(fn))
for (label, _), v in zip(cases, t):
    print(f"{label:40s} {statistics.median(v):8.1f} us")
