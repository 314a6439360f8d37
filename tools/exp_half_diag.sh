#!/bin/bash
# where does the `half` mode (IEEE-half operands + loss scaling) lose its 5-7 % against `fast` (bf16)?  ONE gpurun
# call on one box (run ON the GPU box from the repo root): interleaved per-kernel A/B of the two operand formats, kernel traces of the
# two training steps, and the clock the chip holds inside the bf16 / f16 GEMM and attention loops (ablation build).
#   tools/exp_half_diag.sh [r05]
set -o pipefail
R="${1:-r05}"; O="gpurun_out/${R}_half"; mkdir -p "$O"; export TMPDIR=/tmp
timeout -k 10 400 python3 tools/ab_dtype.py > "$O/ab_dtype.txt" 2> "$O/ab_dtype.err" || { echo "ab_dtype failed"; tail -5 "$O/ab_dtype.err"; exit 1; }
cat "$O/ab_dtype.txt"
for mode in fast half; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_$mode" -- python3 bench.py --precision $mode --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$O/kt_$mode.log" 2>&1 || { echo "kernel trace $mode failed"; tail -5 "$O/kt_$mode.log"; exit 1; }
  python3 tools/summarize_profile.py --round "${R}_$mode" --kt "$O/kt_$mode" --steps 13 --warmup 3 --out "$O" --cmd "python3 bench.py --precision $mode --steps 10 --warmup 3 --no-cpu-baseline --no-extras" > "$O/sum_$mode.log" 2>&1 || { tail -5 "$O/sum_$mode.log"; exit 1; }
  find "$O/kt_$mode" -name "*.csv" -size +8M -delete 2>/dev/null
  tail -1 "$O/kt_$mode.log" | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['value'], j['ms_per_step'])"
done
if [ -f build_exp/libtad_ablation.so ]; then
  for dt in bf16 f16; do
    TAD_LIB=build_exp/libtad_ablation.so timeout -k 10 300 python3 tools/exp_clock.py --dtype $dt --out "$O/clock_$dt.json" > "$O/clock_$dt.log" 2>&1 || { echo "clock $dt failed"; tail -5 "$O/clock_$dt.log"; exit 1; }
    grep -v "^{" "$O/clock_$dt.log"
  done
fi
