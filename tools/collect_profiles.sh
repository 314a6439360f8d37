#!/bin/bash
# One gpurun call's worth of rocprofv3 evidence for profiles/ (run ON the GPU box from the repo root):
#   tools/collect_profiles.sh r03
# bench line, kernel trace + stats of the timed bench command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel trace
# only -- never combined with other trace domains), condensed by tools/summarize_profile.py into profiles/<round>_*.
set -o pipefail
R="${1:-r03}"; O="gpurun_out/${R}_prof"; mkdir -p "$O"; export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" || { echo "bench failed"; tail -5 "$O/bench.err"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$O/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$O/kt.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-profile --no-extras > "$O/fetch.log" 2>&1 || { echo "FETCH pass failed"; tail -5 "$O/fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-profile --no-extras > "$O/write.log" 2>&1 || { echo "WRITE pass failed"; tail -5 "$O/write.log"; exit 1; }
mkdir -p "$O/profiles"
python3 tools/summarize_profile.py --round "$R" --kt "$O/kt" --fetch "$O/fetch" --write "$O/write" --steps 13 --warmup 3 --out "$O/profiles" && cp "$O/bench.json" "$O/profiles/${R}_bench.json"
# keep the merged-back scratch small: the per-dispatch CSVs are large
find "$O/kt" "$O/fetch" "$O/write" -name "*.csv" -size +8M -delete 2>/dev/null
ls -la "$O/profiles"
