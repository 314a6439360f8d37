#!/bin/bash
# PMC passes over the GEMM kernels at the benchmark shapes (run ON the GPU box): where does the operand staging path (TA -> TCP -> L2 -> fabric) spend
# its cycles?   tools/pmc_gemm.sh [tag]   -> gpurun_out/<tag>_pmc_gemm/summary.txt
export TMPDIR=/tmp
O="gpurun_out/${1:-r04}_pmc_gemm"; mkdir -p "$O"
run() {  # name, counters...
  local name="$1"; shift
  timeout -k 10 90 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$O/$name" -- python3 tools/bench_kernels.py --only gemm_nt,gemm_tn --iters 3 > "$O/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$O/$name.log"; return 1; }
}
# (at most two counters of one block per pass: "Request exceeds the capabilities of the hardware to collect" otherwise -- and that failure hangs)
run a TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run a2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum || exit 1
run b TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum || exit 1
run b2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum || exit 1
run b3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum || exit 1
run c TCC_HIT_sum TCC_MISS_sum || exit 1
run c2 TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run d TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE || exit 1
run d2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
order = {}
PASSES = ["a", "a2", "b", "b2", "b3", "c", "c2", "d", "d2"]
for p in PASSES:
    for f in glob.glob(f"{O}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"(gemm_nt_kernel|gemm_tn_kernel)<([^>]*)>", n)
            if not m: continue
            # one line per (kernel instantiation, grid): the benchmark launches every shape with its own grid / template arguments
            k = f"{m.group(1)}<{m.group(2)}> grid {r.get('Grid_Size', '?')}"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][p].add(r["Dispatch_Id"])
lines = []
for k, c in sorted(agg.items()):
    n = {p: max(len(cnt[k][p]), 1) for p in PASSES}
    cyc = c["GRBM_GUI_ACTIVE"] / (n["a"] + n["d"]) / 8.0  # GPU-active cycles per launch (the counter is summed over the 8 XCDs)
    if cyc <= 0: continue
    per = lambda name, p: c[name] / n[p]
    lines.append(f"{k}: launches {n['a']}, {cyc:.3e} GPU-active cycles per launch and XCD; percentages = counter summed over the 256 CUs / (256 x cycles)")
    lines.append(f"   TA busy (sum over 256 TAs) {per('TA_TA_BUSY_sum','a') / cyc / 256 * 100:5.1f} % of cycles; address path stalled by TC {per('TA_ADDR_STALLED_BY_TC_CYCLES_sum','a') / cyc / 256 * 100:5.1f} %, data path stalled by TC {per('TA_DATA_STALLED_BY_TC_CYCLES_sum','a2') / cyc / 256 * 100:5.1f} %; LDS-DMA wave-instructions {per('TA_BUFFER_READ_LDS_WAVEFRONTS_sum','a2'):.3e}")
    lines.append(f"   TCP: busy {per('TCP_GATE_EN1_sum','b') / cyc / 256 * 100:5.1f} %, waiting for L2 data {per('TCP_PENDING_STALL_CYCLES_sum','b') / cyc / 256 * 100:5.1f} %, tag-conflict stalls {per('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum','b3') / cyc / 256 * 100:5.1f} %, TCR->TCP stalls {per('TCP_TCR_TCP_STALL_CYCLES_sum','b3') / cyc / 256 * 100:5.1f} %, TA data stalls {per('TCP_TCP_TA_DATA_STALL_CYCLES_sum','d2') / cyc / 256 * 100:5.1f} %, latency FIFO full {per('TCP_LFIFO_STALL_CYCLES_sum','d2') / cyc / 256 * 100:5.1f} %; read requests to L2 {per('TCP_TCC_READ_REQ_sum','b2'):.3e}, mean L2 read latency {per('TCP_TCC_READ_REQ_LATENCY_sum','b2') / max(per('TCP_TCC_READ_REQ_sum','b2'), 1):.0f} cycles")
    lines.append(f"   L2: requests {per('TCC_REQ_sum','c2'):.3e}, hit {per('TCC_HIT_sum','c'):.3e} miss {per('TCC_MISS_sum','c'):.3e} ({100 * per('TCC_HIT_sum','c') / max(per('TCC_HIT_sum','c') + per('TCC_MISS_sum','c'), 1):.1f} % hits), fabric read requests {per('TCC_EA0_RDREQ_sum','c2'):.3e}")
    lines.append(f"   address translation: UTCL1 hits {per('TCP_UTCL1_TRANSLATION_HIT_sum','d'):.3e} misses {per('TCP_UTCL1_TRANSLATION_MISS_sum','d'):.3e}; TCP cache accesses {per('TCP_TOTAL_CACHE_ACCESSES_sum','d'):.3e}")
open(f"{O}/summary.txt", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
find "$O" -name "*.csv" -size +4M -delete 2>/dev/null; true
