#!/usr/bin/env python3
"""The clock the chip holds INSIDE the MFMA loops (MI355X_MICROARCH.md 'DVFS give-back' item 6).

Needs an ablation build next to the production one (only that build carries the stamps):
    TAD_BUILD_LIB=libtad_ablation.so TAD_BUILD_ABLATION=1 python -m simple_tad_amd.build --force
    TAD_LIB=build_exp/libtad_ablation.so python tools/exp_clock.py --out profiles/r03_clock.json
For each probe: >= 2 s of back-to-back launches on random data, then ONE stamped launch; every workgroup records s_memtime (shader
clock) and s_memrealtime (100 MHz) at the start and the end of its loop (gemm_nt 256x256: around the K loop of each tile; attention
dK/dV: around the tile loop); clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups / tiles.

"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="operand format: the bf16 kernels or their IEEE-half twins (VERDICT r04 item 1)")
a = ap.parse_args()
lib = _lib.load()
dev, bf = "cuda", (torch.float16 if a.dtype == "f16" else torch.bfloat16)
K.set_operand_dtype(bf)
M = 50176
res = {"method": "s_memtime / s_memrealtime x 100 MHz around the loop, one stamped launch after >= %.1f s of back-to-back launches on "
                 "random data; median over workgroups (and tiles)" % a.seconds, "nominal_mhz": 2400, "operands": a.dtype, "device": K.device_info()}


def soak(fn, seconds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return n


# ---- gemm_nt 256x256 (persistent): fc1 forward (K = 768, GELU epilogue) and fc2 forward (K = 3072, f32 residual epilogue)
for name, (n, k, mode) in {"gemm_nt_fc1_K768": (3072, 768, "gelu"), "gemm_nt_qkv_K768": (2304, 768, "plain"),
                           "gemm_nt_dxfc1_K3072": (768, 3072, "plain")}.items():
    x = torch.randn(M, k, device=dev).to(bf)
    w = (torch.randn(n, k, device=dev) * 0.02).to(bf)
    bias = torch.randn(n, device=dev)
    fn = (lambda: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)) if mode == "gelu" else (lambda: K.linear_fwd(x, w, bias))
    launches = soak(fn, a.seconds)
    nwg = 4096
    buf = torch.zeros(nwg * 64 * 32, dtype=torch.int64, device=dev)
    assert lib.tad_linear_debug_stamps(buf.data_ptr()) == 0, lib.tad_last_error_string()
    fn()
    torch.cuda.synchronize()
    lib.tad_linear_debug_stamps(None)
    s = buf.cpu().numpy().reshape(nwg, 64, 32).astype(np.float64)
    used = (s[:, :, 0] > 0) & (s[:, :, 1] > s[:, :, 0])
    d_real = (s[:, :, 1] - s[:, :, 0])[used]          # 10 ns ticks
    d_clk = (s[:, :, 17] - s[:, :, 16])[used]         # shader cycles
    mhz = d_clk / d_real * 100.0
    # time of the launch from the stamps: first tile start to last store acknowledged
    span_us = (s[:, :, 3][used].max() - s[:, :, 0][used].min()) / 100.0
    res[name] = {"tiles": int(used.sum()), "soak_launches": launches, "k_loop_us_median": round(float(np.median(d_real)) / 100.0, 2),
                 "clock_mhz_median": round(float(np.median(mhz)), 1), "clock_mhz_p10": round(float(np.percentile(mhz, 10)), 1),
                 "clock_mhz_p90": round(float(np.percentile(mhz, 90)), 1), "launch_span_us": round(float(span_us), 1),
                 "k_loop_tflops_at_measured_time": round(2.0 * 256 * 256 * k / (float(np.median(d_real)) * 1e-8) * 256 / 1e12, 1)}
    print(name, res[name], flush=True)
    del x, w, buf

# ---- attention dK/dV tile loop
B, N, H = 32, 1568, 12
qkv = torch.randn(B * N, 3 * H * 64, device=dev).to(bf)
ao, lse = K.attn_fwd(qkv, B, N, H, 0.125)
d_ao = torch.randn_like(ao)
fn = lambda: K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125)  # noqa: E731
launches = soak(fn, a.seconds)
nwg = ((N + 127) // 128) * H * B
buf = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
assert lib.tad_attn_debug_stamps(buf.data_ptr()) == 0, lib.tad_last_error_string()
fn()
torch.cuda.synchronize()
lib.tad_attn_debug_stamps(None)
s = buf.cpu().numpy().reshape(nwg, 4).astype(np.float64)
ok = s[:, 2] > s[:, 0]
mhz = (s[ok, 3] - s[ok, 1]) / (s[ok, 2] - s[ok, 0]) * 100.0
res["attn_bwd_dkv_tile_loop"] = {"workgroups": int(ok.sum()), "soak_launches": launches,
                                 "loop_us_median": round(float(np.median(s[ok, 2] - s[ok, 0])) / 100.0, 1),
                                 "clock_mhz_median": round(float(np.median(mhz)), 1), "clock_mhz_p10": round(float(np.percentile(mhz, 10)), 1),
                                 "clock_mhz_p90": round(float(np.percentile(mhz, 90)), 1)}
print("attn_bwd_dkv", res["attn_bwd_dkv_tile_loop"], flush=True)
clocks = [v["clock_mhz_median"] for k, v in res.items() if isinstance(v, dict) and "clock_mhz_median" in v]
res["held_clock_mhz"] = round(float(np.median(clocks)), 1)
import hashlib  # noqa: E402
_h = hashlib.sha256()
_d = os.path.join(ROOT, "simple_tad_amd", "csrc")
for _f in sorted(os.listdir(_d)) + [os.path.join("..", "..", "include", "tad_mi355x.h")]:
    with open(os.path.join(_d, _f), "rb") as _fh:
        _h.update(_fh.read())
res["csrc_sha16"] = _h.hexdigest()[:16]  # bench.py quotes this clock only while the kernel sources still match
res["peak_bf16_tflops_at_held_clock"] = round(256 * 4096 * res["held_clock_mhz"] * 1e6 / 1e12, 1)
print(json.dumps(res))
if a.out:
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
