#!/bin/bash
# PMC pass over the attention kernels at the benchmark shapes (run ON the GPU box): matrix-pipe busy, VALU busy, co-execution, issue stalls.
export TMPDIR=/tmp
O="gpurun_out/${1:-r03}_pmc_attn"; mkdir -p "$O"
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d "$O/p1" -- python3 tools/bench_kernels.py --only attn --iters 3 > "$O/p1.log" 2>&1 || { tail -5 "$O/p1.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$O/p2" -- python3 tools/bench_kernels.py --only attn --iters 3 > "$O/p2.log" 2>&1 || { tail -5 "$O/p2.log"; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
O = sys.argv[1]
def short(n):
    m = re.match(r"(?:void )?tad::(?:op_\w+::)?(\w+)", n); return m.group(1) if m else n[:40]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for p in ("p1", "p2"):
    for f in glob.glob(f"{O}/{p}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "attn" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, p)].add(r["Dispatch_Id"])
lines = []
for k, c in agg.items():
    n1, n2 = max(len(cnt[(k, "p1")]), 1), max(len(cnt[(k, "p2")]), 1)
    wc = c["SQ_WAVE_CYCLES"] / n1
    lines.append(f"{k}: launches {n1}")
    lines.append(f"   matrix pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / n1 / (c['SQ_BUSY_CU_CYCLES'] / n1 * 4 + 1e-9):5.1f} % of CU-busy x 4 SIMD   (MFMA_BUSY {c['SQ_VALU_MFMA_BUSY_CYCLES'] / n1:.3e}, COEXEC {c['SQ_VALU_MFMA_COEXEC_CYCLES'] / n1:.3e} = {100 * c['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(c['SQ_VALU_MFMA_BUSY_CYCLES'], 1):.1f} % of MFMA busy)")
    lines.append(f"   wave cycles {wc:.3e} (quad-cycles): waiting {100 * c['SQ_WAIT_ANY'] / n1 / wc:4.1f} %  issue-stalled {100 * c['SQ_WAIT_INST_ANY'] / n1 / wc:4.1f} %  issuing {100 * c['SQ_ACTIVE_INST_ANY'] / n1 / wc:4.1f} %  (VALU issue {100 * c['SQ_ACTIVE_INST_VALU'] / n1 / wc:4.1f} %)")
    lines.append(f"   per launch: VALU insts {c['SQ_INSTS_VALU'] / n2:.3e} (transcendental {c['SQ_INSTS_VALU_TRANS_F32'] / n2:.3e})  MFMA insts {c['SQ_INSTS_MFMA'] / n2:.3e}  LDS issue {c['SQ_ACTIVE_INST_LDS'] / n2:.3e} LDS-issue stall {c['SQ_WAIT_INST_LDS'] / n2:.3e}  LDS array active {c['SQ_LDS_IDX_ACTIVE'] / n2:.3e}  GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE'] / n2:.3e}")
open(f"{O}/summary.txt", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
find "$O" -name "*.csv" -size +4M -delete 2>/dev/null; true
