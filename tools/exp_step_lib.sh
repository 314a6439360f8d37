#!/bin/bash
# step A/B of two library builds: tools/exp_step_lib.sh libA.so libB.so [rounds]
set -e
A=$1; B=$2; R=${3:-2}
OUT=gpurun_out/steplib; mkdir -p $OUT
for i in $(seq 1 $R); do
  for l in $A $B; do
    n=$(basename $l .so)
    TAD_LIB=$l python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${n}_$i.json 2> $OUT/${n}_$i.err
    python - <<PY
import json
d = json.loads(open("$OUT/${n}_$i.json").read().strip().splitlines()[-1])
print("$n round $i:", d["value"], "clips/s", d["ms_per_step"], "ms", flush=True)
PY
  done
done
