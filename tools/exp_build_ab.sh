#!/bin/bash
# A/B of two builds of the library in alternating processes on one GPU: tools/exp_build_ab.sh LIB_B "command ..." [rounds]
# (LIB_B: an experiment build made with TAD_BUILD_LIB=... TAD_BUILD_DEFINES=... python -m simple_tad_amd.build --force)
libb="$1"; cmd="$2"; rounds="${3:-2}"
for r in $(seq 1 "$rounds"); do
  echo "######## round $r: A = production library"
  bash -c "$cmd" || exit 1
  echo "######## round $r: B = $libb"
  TAD_LIB="$libb" bash -c "$cmd" || exit 1
done
