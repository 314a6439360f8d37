#!/usr/bin/env python3
"""In-kernel timeline of one gemm_nt launch (tad_linear_debug_stamps; needs an ablation build:
TAD_BUILD_ABLATION=1 python -m simple_tad_amd.build --force): per tile K-loop / epilogue / store-drain durations and
the spread of the workgroups' phases.   python tools/exp_timeline.py [--shape fc1] [--config 1,100,1]"""
import argparse, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="fc1")
ap.add_argument("--configs", default="persistent=1")
a = ap.parse_args()
lib = _lib.load()
dev, bf, D, M = "cuda", torch.bfloat16, 768, 50176
n, k, mode = {"fc1": (3072, 768, "gelu"), "proj": (768, 768, "res"), "qkv": (2304, 768, "plain"), "fc2": (768, 3072, "res")}[a.shape]
x = torch.randn(M, k, device=dev).to(bf)
w = (torch.randn(n, k, device=dev) * 0.02).to(bf)
bias = torch.randn(n, device=dev)
res = torch.randn(M, n, device=dev) if mode == "res" else None
if mode == "gelu":
    fn = lambda: K.linear_fwd(x, w, bias, epilogue=1, want_preact=True)
elif mode == "res":
    fn = lambda: K.linear_fwd(x, w, bias, out_dtype=torch.float32, epilogue=2, residual=res)
else:
    fn = lambda: K.linear_fwd(x, w, bias)
for cfg in a.configs.split(";"):
    c = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in cfg.split(","))
    K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, **c})
    for _ in range(3):
        fn()
    nwg = 4096
    buf = torch.zeros(nwg * 64 * 32, dtype=torch.int64, device=dev)
    assert lib.tad_linear_debug_stamps(buf.data_ptr()) == 0, lib.tad_last_error_string()
    fn()
    torch.cuda.synchronize()
    lib.tad_linear_debug_stamps(None)
    s = buf.cpu().numpy().reshape(nwg, 64, 32).astype(np.float64)
    used = s[:, :, 0] > 0
    t0 = s[:, :, 0][used].min()
    s = (s - t0) / 100.0  # us
    kloop = (s[:, :, 1] - s[:, :, 0])[used]
    epi = (s[:, :, 2] - s[:, :, 1])[used]
    drain = (s[:, :, 3] - s[:, :, 2])[used]
    print(f"config {c}: {int(used.sum())} tiles on {int(used.any(axis=1).sum())} workgroups; end {s[:, :, 3][used].max():.1f} us")
    for name, v in (("K loop", kloop), ("epilogue", epi), ("drain", drain)):
        print(f"   {name:9s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    prev = s[:, :, 1]
    for q in range(8):
        if not (s[:, :, 4 + 2 * q][used] > 0).any():
            break
        a1, a2 = s[:, :, 4 + 2 * q], s[:, :, 5 + 2 * q]
        print(f"   chunk {q}: transpose+barrier {np.mean((a1 - prev)[used]):5.2f} us, row pass {np.mean((a2 - a1)[used]):5.2f} us")
        prev = a2
    # per workgroup: when its first K loop starts (launch skew), when its last tile's stores are out, tiles it ran
    nt_wg = used.sum(axis=1)
    live = nt_wg > 0
    first = np.array([s[i, 0, 0] for i in np.where(live)[0]])
    last = np.array([s[i, nt_wg[i] - 1, 3] for i in np.where(live)[0]])
    print(f"   first K loop starts: min {first.min():.1f} p50 {np.percentile(first, 50):.1f} p90 {np.percentile(first, 90):.1f} max {first.max():.1f} us;  workgroup finishes: "
          f"p10 {np.percentile(last, 10):.1f} p50 {np.percentile(last, 50):.1f} p90 {np.percentile(last, 90):.1f} max {last.max():.1f} us;  tiles per workgroup {nt_wg[live].min()}..{nt_wg[live].max()}")
    for ntile in sorted(set(nt_wg[live].tolist())):
        sel = np.where(live)[0][nt_wg[live] == ntile]
        l2 = np.array([s[i, ntile - 1, 3] for i in sel])
        print(f"      workgroups with {ntile} tiles: {len(sel)}, finish p50 {np.percentile(l2, 50):.1f} max {l2.max():.1f} us")
    if c.get('persistent', 1):
        # phase spread: epilogue start times of the 3rd tile of each workgroup, and how many workgroups are inside an epilogue over time
        st = s[:, 2, 1][used[:, 2]]
        print(f"   start of 3rd epilogue: min {st.min():.1f} p25 {np.percentile(st, 25):.1f} p50 {np.percentile(st, 50):.1f} p75 {np.percentile(st, 75):.1f} max {st.max():.1f} us")
    ts = np.arange(0, s[:, :, 3][used].max(), 2.0)
    inside = [(int(((s[:, :, 1] <= t) & (s[:, :, 3] > t) & used).sum())) for t in ts]
    print("   workgroups inside an epilogue every 2 us:", " ".join(str(v) for v in inside[:120]))
