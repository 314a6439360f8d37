#!/usr/bin/env python3
"""The attention forward with 64 query rows per wave (attn_fwd_q64_kernel, tad_attn_tuning("fwd_q64", 1): half the LDS fragment bytes per
score, two waves per SIMD instead of four) against the production kernel (32 rows per wave), interleaved rounds in ONE process on random
data, bit-for-bit equality of out / out_lo / lse checked first:   python tools/exp_attn_q64.py [--rounds 9] [--iters 10]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    for tag, B, N, H in (("ViT-B  B32 H12 N1568", 32, 1568, 12), ("ViT-L  B32 H16 N1568", 32, 1568, 16), ("ViT-S  B32 H6  N1568", 32, 1568, 6),
                         ("MAE enc B32 H16 N392", 32, 392, 16)):
        qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
        qkv[:, :H * 64] = (qkv[:, :H * 64].float() * K.q_prescale_of(0.125)).to(torch.bfloat16)
        res = {}
        for v in (0, 1):
            K.attn_tuning(fwd_q64=v)
            res[v] = K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True, q_prescaled=True)
        same = all(torch.equal(x, y) for x, y in zip(res[0], res[1]))
        t = {0: [], 1: []}
        for _ in range(a.rounds):
            for v in (0, 1):
                K.attn_tuning(fwd_q64=v)
                for _ in range(2):
                    K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True, q_prescaled=True)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True, q_prescaled=True)
                e.record()
                torch.cuda.synchronize()
                t[v].append(1e3 * s.elapsed_time(e) / a.iters)
        K.attn_tuning(fwd_q64=0)
        fl = 4.0 * B * H * N * N * 64
        m0, m1 = statistics.median(t[0]), statistics.median(t[1])
        print(f"{tag}: 32 rows/wave {m0:7.1f} us (min {min(t[0]):7.1f}, {fl / m0 / 1e6:6.0f} TF/s)   64 rows/wave {m1:7.1f} us (min {min(t[1]):7.1f}, "
              f"{fl / m1 / 1e6:6.0f} TF/s)   ratio {m1 / m0:.3f}   bit-identical: {same}", flush=True)
        del qkv, res


if __name__ == "__main__":
    main()
