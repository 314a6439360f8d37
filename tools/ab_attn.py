#!/usr/bin/env python3
"""Interleaved A/B of the attention entry points of several builds of the library in ONE process (cdna_hip_programming.md rule 24):
    python tools/ab_attn.py name=path[:abi3] ...  [--rounds 7] [--iters 10]
Every library is loaded by path with ctypes (RTLD_LOCAL), so equally named symbols do not meet.  `:abi3` marks a round-3 library
(tad_attn_fwd / tad_attn_bwd without the q_prescaled argument; it gets the plain qkv).  Random data, ViT-B shapes (B 32, H 12, N 1568)."""
import argparse
import ctypes as C
import statistics
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--H", type=int, default=12)
ap.add_argument("--N", type=int, default=1568)
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--check-bwd", action="store_true", help="dqkv of every build against the first one's (same forward outputs)")
ap.add_argument("--check", action="store_true", help="forward error of every build against an f64 softmax(q k^T) v of the same 16-bit operands (4 heads)")
a = ap.parse_args()
B, H, N = a.B, a.H, a.N
D = H * 64
dev = "cuda"
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
qkv_p = qkv.clone()
qkv_p[:, :D] = (qkv[:, :D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
dout = torch.randn(B * N, D, device=dev).to(torch.bfloat16)
out = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
lo = torch.empty_like(out)
lse = torch.empty(B, H, N, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(2 * B * H * N, device=dev)
st = torch.cuda.current_stream().cuda_stream
vp, i, f = C.c_void_p, C.c_int, C.c_float


def bind(spec):
    name, path = spec.split("=", 1)
    abi3 = path.endswith(":abi3")
    path = path[:-5] if abi3 else path
    lib = C.CDLL(path)
    if abi3:
        lib.tad_attn_fwd.argtypes = [vp, vp, i, vp, vp, i, i, i, i, f, vp]
        lib.tad_attn_bwd.argtypes = [vp] * 7 + [i, i, i, i, f, vp]
        fwd = lambda: lib.tad_attn_fwd(qkv.data_ptr(), out.data_ptr(), 1, lo.data_ptr(), lse.data_ptr(), B, N, H, 64, 0.125, st)  # noqa: E731
        bwd = lambda: lib.tad_attn_bwd(qkv.data_ptr(), out.data_ptr(), lo.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),  # noqa: E731
                                       delta.data_ptr(), B, N, H, 64, 0.125, st)
    else:
        lib.tad_attn_fwd.argtypes = [vp, vp, i, vp, vp, i, i, i, i, f, i, f, C.c_uint32, vp]
        lib.tad_attn_bwd.argtypes = [vp] * 7 + [i, i, i, i, f, i, f, C.c_uint32, vp]
        fwd = lambda: lib.tad_attn_fwd(qkv_p.data_ptr(), out.data_ptr(), 1, lo.data_ptr(), lse.data_ptr(), B, N, H, 64, 0.125, 1, 0.0, 0, st)  # noqa: E731
        bwd = lambda: lib.tad_attn_bwd(qkv_p.data_ptr(), out.data_ptr(), lo.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),  # noqa: E731
                                       delta.data_ptr(), B, N, H, 64, 0.125, 1, 0.0, 0, st)
    return name, fwd, bwd


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        rc = fn()
    e.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    return s.elapsed_time(e) / a.iters * 1e3  # us


libs = [bind(s) for s in a.libs]
for _, fw, bw in libs:  # warm-up (clocks, caches)
    for _ in range(5):
        fw()
        if not a.fwd_only:
            bw()
torch.cuda.synchronize()
res = {(n, k): [] for n, _, _ in libs for k in ("fwd", "bwd")}
for _ in range(a.rounds):
    for n, fw, bw in libs:
        res[(n, "fwd")].append(timeit(fw))
        res[(n, "bwd")].append(0.0 if a.fwd_only else timeit(bw))
print(f"{'build':16s} {'fwd us med':>10s} {'min':>8s} {'bwd us med':>11s} {'min':>8s}")
for n, _, _ in libs:
    fw, bw = res[(n, "fwd")], res[(n, "bwd")]
    print(f"{n:16s} {statistics.median(fw):10.1f} {min(fw):8.1f} {statistics.median(bw):11.1f} {min(bw):8.1f}")
sys.stdout.flush()

if a.check:
    import math
    q3 = qkv_p.view(B, N, 3, H, 64)
    print(f"{'build':16s} {'fwd rel-L2':>11s} {'max |err| / max |ref|':>22s}   (f64 reference of the same operands, batch 0, heads 0-3)")
    for n, fw, _ in libs:
        fw()
        torch.cuda.synchronize()
        o = out.view(B, N, H, 64)
        num = den = 0.0
        mx = rmx = 0.0
        for h in range(4):
            q, k, v = (q3[0, :, j, h].double() for j in range(3))
            ref = torch.softmax((q @ k.t()) * math.log(2.0), dim=-1) @ v
            d = o[0, :, h].double() - ref
            num += float((d * d).sum()); den += float((ref * ref).sum())
            mx = max(mx, float(d.abs().max())); rmx = max(rmx, float(ref.abs().max()))
        print(f"{n:16s} {math.sqrt(num / den):11.2e} {mx / rmx:22.2e}")

if a.check_bwd:
    ref = None
    for n, fw, bw in libs:
        libs[0][1]()  # (the first build's forward for all: same out / lse)
        dqkv.zero_()
        bw()
        torch.cuda.synchronize()
        if ref is None:
            ref = dqkv.clone()
            print(f"{n:16s} dqkv reference, |dqkv| max {float(ref.float().abs().max()):.3e}")
        else:
            d = (dqkv.float() - ref.float()).abs()
            print(f"{n:16s} dqkv bit-identical to {libs[0][0]}: {torch.equal(dqkv, ref)}, max |diff| {float(d.max()):.3e}")
