"""Software-pipelined dK/dV kernel (tad_attn_tuning("dkv_pipe", 1)) against the production one: results and time.

    python tools/exp_dkv_pipe.py [--B 32 --N 1568 --H 12] [--rounds 3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--N", type=int, default=1568)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
K.set_operand_dtype(dt)
B, N, H = a.B, a.N, a.H
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(dt)
ao, lse, lo = K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True)
d_ao = torch.randn_like(ao)


def run(pipe):
    K.attn_tuning(dkv_pipe=pipe)
    return K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo)


ref = run(0).float().view(B, N, 3, H, 64)
got = run(1).float().view(B, N, 3, H, 64)
torch.cuda.synchronize()
for i, name in enumerate("qkv"):
    r, g = ref[:, :, i], got[:, :, i]
    print(name, "rel-l2 pipe vs production: %.3e" % ((g - r).norm() / r.norm()).item(), "max abs %.3e" % (g - r).abs().max().item(),
          "finite", bool(torch.isfinite(g).all()))
# per key position: where do they differ?
dk = (got[:, :, 1] - ref[:, :, 1]).abs().amax(dim=(0, 2, 3))
bad = (dk > 0.05 * ref[:, :, 1].abs().max()).nonzero().flatten()
print("keys with large dK deviation:", bad[:20].tolist(), "count", bad.numel())


def timed(pipe, n=20):
    K.attn_tuning(dkv_pipe=pipe)
    for _ in range(3):
        K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for r in range(a.rounds):
    print("round", r, "production %.1f us" % timed(0), "pipelined (4 waves) %.1f us" % timed(1), "pipelined (2 waves) %.1f us" % timed(2),
          "(dQ + dK/dV per call)", flush=True)
K.attn_tuning(dkv_pipe=0)
