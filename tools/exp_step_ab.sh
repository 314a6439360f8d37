#!/bin/bash
# Step-level A/B of two builds of the library: alternating bench.py processes (no extras, no CPU baseline), N rounds; prints clips/s and ms/step per run.
#   tools/exp_step_ab.sh LIB_B [rounds] [steps]      (LIB_A = the production library; LIB_B built with TAD_BUILD_LIB=... TAD_BUILD_DEFINES=...)
libb="$1"; rounds="${2:-4}"; steps="${3:-30}"
for r in $(seq 1 "$rounds"); do
  for lib in libtad_mi355x.so "$libb"; do
    echo -n "$lib "
    TAD_LIB="$PWD/simple_tad_amd/$lib" timeout -k 10 200 python bench.py --steps "$steps" --no-extras --no-cpu-baseline 2>&1 | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
