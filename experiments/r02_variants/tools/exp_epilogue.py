#!/usr/bin/env python3
"""Experiment: gemm_nt scheduling (per-tile grid vs persistent workgroups with staggered starts) at the real Linear shapes.
    python tools/exp_epilogue.py [--configs "0,100,1;1,0,1;1,100,1;1,100,8;1,50,8"]    key=value knobs of kernels.linear_tuning"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K, _lib
from tools.bench_kernels import timeit

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="persistent=1,split_tail=0;persistent=1,split_tail=2;persistent=1,split_tail=1")
ap.add_argument("--M", type=int, default=50176)
a = ap.parse_args()
lib = _lib.load()
dev, bf, D, M = "cuda", torch.bfloat16, 768, a.M
shapes = [("qkv", 3 * D, D, "plain"), ("proj", D, D, "res"), ("fc1", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "res"),
          ("dXqkv", D, 3 * D, "plain_f32"), ("dXproj", D, D, "plain"), ("dXfc2", 4 * D, D, "dgelu"), ("dXfc1", D, 4 * D, "plain")]
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",")) for c in a.configs.split(";")]
for i, c in enumerate(cfgs):
    print(f"config {i}: {c}")
print("shape    " + "".join(f"{('cfg' + str(i)):>10s}" for i in range(len(cfgs))) + "   (us)")
tot = [0.0] * len(cfgs)
for name, n, k, mode in shapes:
    x = torch.randn(M, k, device=dev).to(bf)
    w = (torch.randn(n, k, device=dev) * 0.02).to(bf)
    bias = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev) if mode == "res" else None
    h = torch.randn(M, n, device=dev).to(bf) if mode == "dgelu" else None
    if mode == "gelu":
        fn = lambda: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)
    elif mode == "res":
        fn = lambda: K.linear_fwd(x, w, bias, out_dtype=torch.float32, epilogue=2, residual=res)
    elif mode == "dgelu":
        fn = lambda: K.linear_bwd_input(x, w, gelu_preact=h)
    elif mode == "plain_f32":
        fn = lambda: K.linear_fwd(x, w, None, out_dtype=torch.float32)
    else:
        fn = lambda: K.linear_fwd(x, w, bias)
    line = f"{name:8s} "
    for i, c in enumerate(cfgs):
        K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, **c})
        us = timeit(fn, 30) * 1000
        tot[i] += us
        line += f"{us:10.1f}"
    print(line, flush=True)
    del x, w, res, h
print("sum      " + "".join(f"{t:10.1f}" for t in tot))
