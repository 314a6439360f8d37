#!/bin/bash
# The whole profiles/<round>_* set from ONE gpurun call on the final tree (run ON the GPU box from the repo root):
#   tools/collect_final.sh r05
# Order matters: the in-kernel clock (ablation build, made in the build container beforehand) and the rocprofv3 summaries are
# installed into profiles/ BEFORE `python bench.py` runs, so the bench line's `from_profiles` entries are current for the sources
# it runs on.  Everything is written under gpurun_out/<round>_final/profiles/ too (what travels back).
set -o pipefail
R="${1:-r06}"; O="gpurun_out/${R}_final"; P="$O/profiles"; mkdir -p "$P"; export TMPDIR=/tmp
if [ -f build_exp/libtad_ablation.so ]; then
  TAD_LIB=build_exp/libtad_ablation.so timeout -k 10 300 python3 tools/exp_clock.py --out "$P/${R}_clock.json" > "$O/clock.log" 2>&1 || { echo "clock failed"; tail -5 "$O/clock.log"; exit 1; }
  cp "$P/${R}_clock.json" profiles/
fi
if [ -f build_exp/libtad_ablation.so ]; then  # the IEEE-half twins' clock: not read by bench.py, kept beside the bf16 one
  TAD_LIB=build_exp/libtad_ablation.so timeout -k 10 300 python3 tools/exp_clock.py --dtype f16 --out "$P/${R}_clock_f16.json" > "$O/clock_f16.log" 2>&1 && cp "$P/${R}_clock_f16.json" profiles/ || echo "f16 clock failed"
fi
echo "[collect_final] clock done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$O/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$O/kt.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_half" -- python3 bench.py --precision half --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$O/kt_half.log" 2>&1 \
  && python3 tools/summarize_profile.py --round "${R}_half" --kt "$O/kt_half" --steps 13 --warmup 3 --out "$O" --cmd "python3 bench.py --precision half --steps 10 --warmup 3 --no-cpu-baseline --no-extras" > "$O/summarize_half.log" 2>&1 \
  && cp "$O/${R}_half_kernel_stats.txt" "profiles/${R}_kernel_stats_half.txt" && cp "$O/${R}_half_kernel_stats.txt" "$P/${R}_kernel_stats_half.txt" || echo "half kernel trace failed"
find "$O/kt_half" -name "*.csv" -size +8M -delete 2>/dev/null
echo "[collect_final] kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-profile --no-extras > "$O/fetch.log" 2>&1 || { echo "FETCH pass failed"; tail -5 "$O/fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-profile --no-extras > "$O/write.log" 2>&1 || { echo "WRITE pass failed"; tail -5 "$O/write.log"; exit 1; }
echo "[collect_final] PMC passes done"
python3 tools/summarize_profile.py --round "$R" --kt "$O/kt" --fetch "$O/fetch" --write "$O/write" --steps 13 --warmup 3 --out "$P" > "$O/summarize.log" 2>&1 || { tail -5 "$O/summarize.log"; exit 1; }
cp "$P/${R}_summary.json" "$P/${R}_kernel_stats.txt" profiles/
find "$O/kt" "$O/fetch" "$O/write" -name "*.csv" -size +8M -delete 2>/dev/null
timeout -k 10 900 python3 bench.py > "$P/${R}_bench.json" 2> "$O/bench.err" || { echo "bench failed"; tail -5 "$O/bench.err"; exit 1; }
echo "[collect_final] bench done"
if tools/pmc_attention.sh "$R" > "$O/pmc_attn.log" 2>&1; then cp "gpurun_out/${R}_pmc_attn/summary.txt" "$P/${R}_pmc_attention.txt"; else echo "pmc_attention failed"; tail -5 "$O/pmc_attn.log"; fi
ls -la "$P"
python3 - "$P/${R}_bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "frac_of_bf16_mfma_roofline")})
print("roofline", j.get("roofline"))
for k in ("engine_loop", "fwd_only", "half", "precise", "mae_pretrain", "vit_small", "vit_large", "torch_route"):
    o = j.get(k)
    if isinstance(o, dict):
        print(k, {kk: vv for kk, vv in o.items() if not isinstance(vv, (dict, list))})
print("cpu_baseline", j.get("cpu_baseline"))
print("from_profiles", j.get("from_profiles"))
PY
