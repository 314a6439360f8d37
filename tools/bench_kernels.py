#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the real ViT-B/16 16x224^2 B=32 shapes (M = 50176).
Times each C-ABI entry with HIP events on random data (never zero-filled operands).

    python tools/bench_kernels.py [--iters 20] [--only gemm_nt,attn] [--B 32]
Environment: TAD_GEMM_NT_VARIANT / TAD_GEMM_TN_VARIANT select tile configurations (see csrc/gemm.hip).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--D", type=int, default=768)
    a = ap.parse_args()
    only = set(x for x in a.only.split(",") if x)
    dev = "cuda"
    B, N, D, H = a.B, 1568, a.D, a.D // 64
    M = B * N
    bf = torch.bfloat16

    def rnd(*shape, dtype=bf, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    def want(name):
        return not only or any(name.startswith(o) for o in only)

    rows = []
    if want("gemm_nt"):
        for (name, n, k, kw) in [("qkv  fwd  N=2304 K=768", 3 * D, D, {}), ("proj fwd  N=768  K=768  +res", D, D, {"res": True}),
                                 ("fc1  fwd  N=3072 K=768  gelu", 4 * D, D, {"gelu": True}), ("fc2  fwd  N=768  K=3072 +res", D, 4 * D, {"res": True}),
                                 ("dX   N=768  K=2304", D, 3 * D, {}), ("dX   N=3072 K=768 dgelu", 4 * D, D, {"dgelu": True})]:
            x, w = rnd(M, k), rnd(n, k, scale=0.02)
            bias = torch.randn(n, device=dev)
            res = torch.randn(M, n, device=dev) if kw.get("res") else None
            h = rnd(M, n) if kw.get("dgelu") else None
            if kw.get("dgelu"):
                fn = lambda: K.linear_bwd_input(x, w, gelu_preact=h)  # noqa: E731  (w plays the role of W^T [K_out=n, N_in=k])
            elif kw.get("gelu"):
                fn = lambda: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)  # noqa: E731
            elif kw.get("res"):
                fn = lambda: K.linear_fwd(x, w, bias, out_dtype=torch.float32, epilogue=2, residual=res)  # noqa: E731
            else:
                fn = lambda: K.linear_fwd(x, w, bias)  # noqa: E731
            ms = timeit(fn, a.iters)
            rows.append(("gemm_nt " + name, ms, 2.0 * M * n * k / ms / 1e9))
            del x, w, res, h
    if want("gemm_tn"):
        for (name, n, k) in [("dWqkv N=2304 K=768", 3 * D, D), ("dWproj N=768 K=768", D, D), ("dWfc1 N=3072 K=768", 4 * D, D),
                             ("dWfc2 N=768 K=3072", D, 4 * D)]:
            dy, x = rnd(M, n), rnd(M, k)
            ms = timeit(lambda: K.linear_bwd_weight(dy, x, want_bias=False), a.iters)
            rows.append(("gemm_tn " + name, ms, 2.0 * M * n * k / ms / 1e9))
            ms2 = timeit(lambda: K.colsum_bf16(dy), a.iters)
            rows.append(("colsum  " + name.split()[1], ms2, 0.0))
            del dy, x
    if want("attn"):
        qkv = rnd(M, 3 * D)
        dout = rnd(M, D)
        # production contract: the q third carries scale * log2e (tad_linear_fwd_qkv's q_prescale); plain q = flash-attn's contract
        qkv_p = qkv.clone()
        qkv_p[:, :D] = (qkv[:, :D].float() * K.q_prescale_of(0.125)).to(qkv.dtype)
        for tag, t, pre in (("q pre-scaled", qkv_p, True), ("plain q", qkv, False)):
            out, lse = K.attn_fwd(t, B, N, H, 0.125, q_prescaled=pre)
            ms = timeit(lambda: K.attn_fwd(t, B, N, H, 0.125, q_prescaled=pre), a.iters)
            rows.append((f"attn_fwd [{tag}]", ms, 4.0 * B * H * N * N * 64 / ms / 1e9))
            out, lse, lo = K.attn_fwd(t, B, N, H, 0.125, want_lo=True, q_prescaled=pre)
            ms = timeit(lambda: K.attn_fwd(t, B, N, H, 0.125, want_lo=True, q_prescaled=pre), a.iters)
            rows.append((f"attn_fwd + out_lo [{tag}]", ms, 4.0 * B * H * N * N * 64 / ms / 1e9))
            ms = timeit(lambda: K.attn_bwd(t, out, dout, lse, B, N, H, 0.125, out_lo=lo, q_prescaled=pre), a.iters)
            rows.append((f"attn_bwd (alg. 2x fwd flops) [{tag}]", ms, 8.0 * B * H * N * N * 64 / ms / 1e9))
    if want("ln"):
        x = torch.randn(M, D, device=dev)
        g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
        ms = timeit(lambda: K.layernorm_fwd(x, g, b, 1e-6), a.iters)
        rows.append((f"layernorm_fwd ({6.0 * M * D / ms / 1e6:.0f} GB/s)", ms, 0.0))
        dy = rnd(M, D)
        ms = timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=x, want_bf16=True, want_colsum=True), a.iters)
        rows.append((f"layernorm_bwd ({16.0 * M * D / ms / 1e6:.0f} GB/s)", ms, 0.0))
        ms = timeit(lambda: K.cast_bf16(x), a.iters)
        rows.append((f"cast ({6.0 * M * D / ms / 1e6:.0f} GB/s)", ms, 0.0))
    if want("patch"):
        xv = torch.randn(B, 3, 16, 224, 224, device=dev)
        w = rnd(D, 1536, scale=0.02)
        pos = torch.randn(N, D, device=dev)
        bias = torch.zeros(D, device=dev)
        ms = timeit(lambda: K.patch_embed_fwd(xv, w, bias, pos, 2, 16), a.iters)
        rows.append(("patch_embed_fwd", ms, 2.0 * M * D * 1536 / ms / 1e9))
        ms = timeit(lambda: K.patch_embed_fwd_implicit(xv, w, bias, pos, 2, 16), a.iters)
        rows.append(("patch_embed_fwd_implicit (no patch matrix; eval / no_grad)", ms, 2.0 * M * D * 1536 / ms / 1e9))
        ms = timeit(lambda: K.im2col_tubelets(xv, 2, 16), a.iters)
        rows.append((f"  im2col alone ({(4.0 * xv.numel() + 2.0 * M * 1536) / ms / 1e6:.0f} GB/s)", ms, 0.0))
        fr = torch.randint(0, 256, (B, 16, 224, 224, 3), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: K.im2col_tubelets_u8(fr, 2, 16, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), a.iters)
        rows.append((f"  im2col from uint8 frames ({(1.0 * fr.numel() + 2.0 * M * 1536) / ms / 1e6:.0f} GB/s)", ms, 0.0))
    if want("adamw"):
        n = 86_228_738 // 4096 * 4096 + 4096 * 152
        bufs = [torch.randn(n, device=dev) for _ in range(2)] + [torch.zeros(n, device=dev) for _ in range(2)]
        mirror = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        cg = torch.zeros(n // 4096, dtype=torch.uint8, device=dev)
        part = torch.zeros(n // 4096, device=dev)
        ms = timeit(lambda: K.adamw_step(bufs[0], bufs[1], bufs[2], bufs[3], cg, [1e-3], [0.05], [1], 0.9, 0.999, 1e-8, param_bf16=mirror,
                                         sumsq_partials=part), a.iters)
        rows.append((f"adamw_step 86M params (+bf16 mirror, +norm) ({30.0 * n / ms / 1e6:.0f} GB/s)", ms, 0.0))
    print(f"{'kernel':48s} {'ms':>9s} {'TFLOP/s':>9s}")
    for name, ms, tf in rows:
        print(f"{name:48s} {ms:9.3f} {tf:9.1f}")


if __name__ == "__main__":
    main()
