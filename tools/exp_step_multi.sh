#!/bin/bash
# Step-level comparison of several builds of the library: `rounds` passes over "production + every listed library", one bench.py process each.
#   tools/exp_step_multi.sh ROUNDS STEPS libtad_x.so libtad_y.so ...
rounds="$1"; steps="$2"; shift 2
for r in $(seq 1 "$rounds"); do
  for lib in libtad_mi355x.so "$@"; do
    echo -n "$lib "
    TAD_LIB="$PWD/simple_tad_amd/$lib" timeout -k 10 200 python bench.py --steps "$steps" --no-extras --no-cpu-baseline 2>&1 | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
