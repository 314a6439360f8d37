#!/bin/bash
# Step A/B of one environment knob of the library, alternated in one gpurun call:  tools/exp_step_knob.sh TAD_GEMM_TAIL_192 0 1 [rounds]
# (each leg is a fresh process: python bench.py --no-extras --no-cpu-baseline; the knob's env var is read when the library loads)
set -e
VAR=$1; A=$2; B=$3; R=${4:-2}
OUT=gpurun_out/knob_${VAR}; mkdir -p $OUT
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/run_${v}_$i.json 2> $OUT/run_${v}_$i.err
    python - <<PY
import json
d = json.loads(open("$OUT/run_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v round $i:", d["value"], "clips/s", d["ms_per_step"], "ms", flush=True)
PY
  done
done
