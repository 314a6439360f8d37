#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>/*/…) into the small text/JSON summaries committed under profiles/.

    python tools/summarize_profile.py --round r01 --kt gpurun_out/r01_kt --fetch gpurun_out/r01_fetch --write gpurun_out/r01_write \
           --steps 13 --pmc-steps 3
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"tad::(?:op_bf16::|(op_f16::))?(\w+)", name)  # (the bf16 / half compilation passes live in inline namespaces)
    if m:
        return "tad::" + (m.group(1) or "") + m.group(2)
    m = re.match(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
    return ("torch::" + m.group(1)) if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--steps", type=int, default=13, help="steps+warmup of the kernel-trace run")
    ap.add_argument("--warmup", type=int, default=3, help="warm-up steps of the kernel-trace run")
    ap.add_argument("--pmc-steps", type=int, default=3)
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--cmd", default="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras", help="the profiled command (header line)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "simple_tad_amd", "csrc")
    for fn in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "tad_mi355x.h")]:
        with open(os.path.join(d, fn), "rb") as fh:
            h.update(fh.read())
    # the kernel sources these profiles were taken on: bench.py reports figures read from here only while the sources still match
    summary = {"round": a.round, "csrc_sha16": h.hexdigest()[:16]}
    if a.kt:
        # Per-dispatch begin/end from the kernel trace (the --stats table of the same run is kept for reference: it is built from
        # the same dispatches, but one first-use dispatch can carry a multi-millisecond outlier there -- see profiles/README.md).
        tf = glob.glob(os.path.join(a.kt, "*", "*kernel_trace.csv"))
        agg = collections.OrderedDict()
        total = 0.0
        if tf:
            nt = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                        for r in csv.DictReader(open(tf[0])) if short(r["Kernel_Name"]) == "tad::gemm_nt_kernel")
            if nt:  # the timed steps only (what bench.py's live HIP-event figure covers): drop the warm-up steps' launches
                skip = len(nt) * a.warmup // a.steps
                timed = [d for _, d in nt[skip:]]
                summary["gemm_nt_timed_steps"] = {"calls": len(timed), "avg_launch_us": sum(timed) / len(timed) / 1e3,
                                                  "ms_per_step": sum(timed) / 1e6 / (a.steps - a.warmup)}
            for r in csv.DictReader(open(tf[0])):
                k = short(r["Kernel_Name"])
                d = agg.setdefault(k, {"calls": 0, "total_ns": 0.0})
                d["calls"] += 1
                d["total_ns"] += float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                total += float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            # Idle time of the device between kernels over the timed steps: span of the dispatches minus the union of their intervals, and
            # which kernel the gaps sit in front of (a gap = the device has nothing running: launch latency, host-side bubbles, dependent-
            # dispatch drain)
            ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tf[0])))
            ev = ev[len(ev) * a.warmup // a.steps:]
            if ev:
                gaps = collections.defaultdict(lambda: [0, 0.0])
                busy_end, idle = ev[0][1], 0.0
                for s0, e0, k in ev[1:]:
                    if s0 > busy_end:
                        idle += s0 - busy_end
                        gaps[k][0] += 1
                        gaps[k][1] += s0 - busy_end
                    busy_end = max(busy_end, e0)
                nst = a.steps - a.warmup
                span = ev[-1][1] - ev[0][0]
                summary["timeline_timed_steps"] = {
                    "span_ms_per_step": span / 1e6 / nst, "idle_ms_per_step": idle / 1e6 / nst, "dispatches_per_step": len(ev) / nst,
                    "gaps_in_front_of": {k: {"per_step": v[0] / nst, "ms_per_step": v[1] / 1e6 / nst, "avg_us": v[1] / v[0] / 1e3}
                                         for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]}}
        else:
            f = glob.glob(os.path.join(a.kt, "*", "*kernel_stats.csv"))[0]
            for r in csv.DictReader(open(f)):
                k = short(r["Name"])
                d = agg.setdefault(k, {"calls": 0, "total_ns": 0.0})
                d["calls"] += int(r["Calls"])
                d["total_ns"] += float(r["TotalDurationNs"])
                total += float(r["TotalDurationNs"])
        sf = glob.glob(os.path.join(a.kt, "*", "*kernel_stats.csv"))
        if sf:
            st = {"calls": 0, "total_ns": 0.0, "max_ns": 0.0}
            for r in csv.DictReader(open(sf[0])):
                if short(r["Name"]) == "tad::gemm_nt_kernel":
                    st["calls"] += int(r["Calls"])
                    st["total_ns"] += float(r["TotalDurationNs"])
                    st["max_ns"] = max(st["max_ns"], float(r["MaxNs"]))
            if st["calls"]:
                summary["gemm_nt_stats_table"] = {"calls": st["calls"], "avg_launch_us": st["total_ns"] / st["calls"] / 1e3,
                                                  "max_launch_us": st["max_ns"] / 1e3,
                                                  "avg_launch_us_without_max": (st["total_ns"] - st["max_ns"]) / (st["calls"] - 1) / 1e3}
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["total_ns"])
        lines = [f"# rocprofv3 --kernel-trace --stats -- {a.cmd}   ({a.steps} steps incl. warm-up)",
                 f"# total kernel time {total / 1e6:.1f} ms = {total / 1e6 / a.steps:.2f} ms/step",
                 f"{'kernel':44s} {'calls':>7s} {'ms/step':>9s} {'avg us':>9s} {'share':>7s}"]
        for k, d in rows[:40]:
            lines.append(f"{k:44s} {d['calls']:7d} {d['total_ns'] / 1e6 / a.steps:9.3f} {d['total_ns'] / d['calls'] / 1e3:9.1f} {100 * d['total_ns'] / total:6.1f}%")
        open(os.path.join(a.out, f"{a.round}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
        g = agg.get("tad::gemm_nt_kernel")
        if g:
            summary["gemm_nt"] = {"calls": g["calls"], "avg_launch_us": g["total_ns"] / g["calls"] / 1e3,
                                  "ms_per_step": g["total_ns"] / 1e6 / a.steps}
        print("\n".join(lines[:14]))
    for tag, path, counter in (("fetch", a.fetch, "FETCH_SIZE"), ("write", a.write, "WRITE_SIZE")):
        if not path:
            continue
        f = glob.glob(os.path.join(path, "*", "*counter_collection.csv"))[0]
        per = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"])
            per[k][1].add(r["Dispatch_Id"])
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of wide streaming reads -> x2
        # (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-byte stores.
        corr = 2.0 if counter == "FETCH_SIZE" else 1.0
        out = {k: {"launches": len(v[1]), "bytes_per_launch": v[0] * 1024.0 * corr / max(len(v[1]), 1)} for k, v in per.items()}
        summary[tag] = {k: out[k] for k in sorted(out, key=lambda k: -out[k]["bytes_per_launch"] * out[k]["launches"])[:12]}
    if "fetch" in summary and "write" in summary and "tad::gemm_nt_kernel" in summary["fetch"]:
        fb = summary["fetch"]["tad::gemm_nt_kernel"]["bytes_per_launch"]
        wb = summary["write"].get("tad::gemm_nt_kernel", {"bytes_per_launch": 0})["bytes_per_launch"]
        summary["gemm_nt_traffic_bytes_per_launch"] = fb + wb
        summary["gemm_nt_traffic_note"] = "FETCH_SIZE KiB x1024 x2 (gfx950 half-count correction) + WRITE_SIZE KiB x1024, averaged over all gemm_nt launches of bench.py"
    json.dump(summary, open(os.path.join(a.out, f"{a.round}_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k.startswith("gemm_nt") or k.startswith("timeline")}, indent=1))


if __name__ == "__main__":
    main()
