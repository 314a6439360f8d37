#!/usr/bin/env python3
"""Workload of tools/pmc_gemm.sh: the eight Linear launches of a ViT-B block exactly as the model issues them (planned launches, in-model
epilogues and output types) and the weight gradients (the qkv + proj pair, fc1, fc2), each called `--iters` times in a fixed order, so that
a PMC pass (one rocprofv3 run per counter set) can attribute every dispatch to a shape by its position in the dispatch sequence.
`--manifest f.json` records, per shape, how many gemm kernel launches one call makes and its algorithmic bytes / flops.

    python tools/pmc_gemm_shapes.py --iters 3 --manifest gpurun_out/x/manifest.json [--D 768] [--B 32]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--D", type=int, default=768)
    ap.add_argument("--manifest", default="")
    a = ap.parse_args()
    dev, bf = "cuda", torch.bfloat16
    D, M = a.D, a.B * 1568

    def rnd(*shape, dtype=bf, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    shapes = []  # (name, kind, N, K, fn, algorithmic bytes)
    # forward / input-gradient Linears as ops.BlockFn issues them (out dtype, epilogue)
    for name, n, k, kind in (("qkv fwd", 3 * D, D, "qkv"), ("proj fwd +res", D, D, "res"), ("fc1 fwd gelu", 4 * D, D, "gelu"), ("fc2 fwd +res", D, 4 * D, "res"),
                             ("dX qkv f32", D, 3 * D, "dx32"), ("dX fc2 dgelu", 4 * D, D, "dgelu"), ("dX fc1 f32", D, 4 * D, "dx32"), ("dX proj", D, D, "dx16")):
        x, w = rnd(M, k), rnd(n, k, scale=0.02)
        bias = torch.randn(n, device=dev)
        if kind == "qkv":
            qb, vb = torch.randn(n // 3, device=dev), torch.randn(n // 3, device=dev)
            fn = lambda x=x, w=w, qb=qb, vb=vb: K.linear_fwd_qkv(x, w, qb, vb, q_prescale=K.q_prescale_of(0.125))  # noqa: E731
            nbytes = 2 * (M * k + n * k) + 2 * M * n
        elif kind == "res":
            res = torch.randn(M, n, device=dev)
            fn = lambda x=x, w=w, bias=bias, res=res: K.linear_fwd(x, w, bias, out_dtype=torch.float32, epilogue=2, residual=res)  # noqa: E731
            nbytes = 2 * (M * k + n * k) + 4 * M * n + 4 * M * n
        elif kind == "gelu":
            fn = lambda x=x, w=w, bias=bias: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)  # noqa: E731
            nbytes = 2 * (M * k + n * k) + 2 * M * n + 2 * M * n
        elif kind == "dgelu":
            h = rnd(M, n)
            fn = lambda x=x, w=w, h=h: K.linear_bwd_input(x, w, gelu_preact=h)  # noqa: E731   (w plays W^T [n, k])
            nbytes = 2 * (M * k + n * k) + 2 * M * n + 2 * M * n
        elif kind == "dx32":
            fn = lambda x=x, w=w: K.linear_bwd_input(x, w, out_dtype=torch.float32)  # noqa: E731
            nbytes = 2 * (M * k + n * k) + 4 * M * n
        else:
            fn = lambda x=x, w=w: K.linear_bwd_input(x, w)  # noqa: E731
            nbytes = 2 * (M * k + n * k) + 2 * M * n
        shapes.append({"name": name, "kernel": "gemm_nt", "N": n, "K": k, "fn": fn, "bytes": nbytes, "a_bytes": 2 * M * k, "w_bytes": 2 * n * k,
                       "flops": 2.0 * M * n * k})
    dy3, dy1, dy4, x1, x4 = rnd(M, 3 * D), rnd(M, D), rnd(M, 4 * D), rnd(M, D), rnd(M, 4 * D)
    dW3, dW1 = torch.zeros(3 * D, D, device=dev), torch.zeros(D, D, device=dev)
    dqb, dvb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    x1b = rnd(M, D)
    shapes.append({"name": "dW fc1", "kernel": "gemm_tn", "N": 4 * D, "K": D, "fn": lambda: K.linear_bwd_weight(dy4, x1, want_bias=True),
                   "bytes": 2 * M * 5 * D + 4 * 4 * D * D, "flops": 2.0 * M * 4 * D * D})
    shapes.append({"name": "dW fc2", "kernel": "gemm_tn", "N": D, "K": 4 * D, "fn": lambda: K.linear_bwd_weight(dy1, x4, want_bias=True),
                   "bytes": 2 * M * 5 * D + 4 * 4 * D * D, "flops": 2.0 * M * 4 * D * D})
    # the qkv and proj weight gradients of a Block: ONE launch (tad_linear_bwd_weight_pair)
    shapes.append({"name": "dW qkv+proj", "kernel": "gemm_tn", "N": 4 * D, "K": D,
                   "fn": lambda: K.linear_bwd_weight_pair(dy3, x1, dW3, dqb, dvb, dy1, x1b, dW1, False),
                   "bytes": 2 * M * 6 * D + 4 * 4 * D * D, "flops": 2.0 * M * 4 * D * D})
    man = []
    for s in shapes:
        s["fn"]()  # (allocations, workspace growth: outside the counted calls of this shape? no -- every call is counted alike; see `calls`)
        torch.cuda.synchronize()
        n0 = K.linear_kernel_launches()
        for _ in range(a.iters):
            s["fn"]()
        torch.cuda.synchronize()
        nl = (K.linear_kernel_launches() - n0) // a.iters if s["kernel"] == "gemm_nt" else 1
        man.append({k: v for k, v in s.items() if k != "fn"} | {"calls": a.iters + 1, "gemm_launches_per_call": int(nl)})
        print(f"{s['name']:16s} {s['kernel']}  N={s['N']:5d} K={s['K']:5d}  launches/call {nl}  algorithmic {s['bytes'] / 1e6:8.1f} MB", flush=True)
    if a.manifest:
        os.makedirs(os.path.dirname(os.path.abspath(a.manifest)), exist_ok=True)
        json.dump({"M": M, "D": D, "iters": a.iters, "shapes": man}, open(a.manifest, "w"), indent=1)


if __name__ == "__main__":
    main()
