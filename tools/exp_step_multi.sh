#!/bin/bash
# Step-level A/B of several builds of the library on ONE box: `rounds` passes over the listed libraries, one bench.py process each
# (no extras, no CPU baseline), alternating so that box drift hits every build alike (cdna_hip_programming.md rule 24).
#   tools/exp_step_multi.sh ROUNDS STEPS [prod] [build_exp/libtad_x.so ...] [KNOB=VALUE:prod ...]
# `prod` = simple_tad_amd/libtad_mi355x.so; a path = an experiment build (TAD_BUILD_LIB=libtad_x.so TAD_BUILD_DEFINES="..." python -m
# simple_tad_amd.build --force -> build_exp/libtad_x.so); `ENV=V:lib` runs that library with one environment knob set (TAD_GEMM_...).
# Prints "<name> <clips/s> <ms/step>" per run and the per-build medians at the end.
rounds="$1"; steps="$2"; shift 2
out=$(mktemp)
for r in $(seq 1 "$rounds"); do
  for spec in "$@"; do
    envset=""; lib="$spec"
    case "$spec" in *=*:*) envset="${spec%%:*}"; lib="${spec#*:}";; esac
    [ "$lib" = prod ] && path="$PWD/simple_tad_amd/libtad_mi355x.so" || path="$PWD/$lib"
    [ -f "$path" ] || { echo "no such library: $path"; exit 1; }
    v=$(env ${envset:+"$envset"} TAD_LIB="$path" timeout -k 10 240 python bench.py --steps "$steps" --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || { echo "$spec failed"; exit 1; }
    echo "$spec $v" | tee -a "$out"
  done
done
python - "$out" <<'PY'
import sys, statistics, collections
d = collections.OrderedDict()
for ln in open(sys.argv[1]):
    n, v, ms = ln.split()
    d.setdefault(n, []).append((float(v), float(ms)))
base = None
for n, vs in d.items():
    med = statistics.median(v for v, _ in vs)
    base = base or med
    print(f"median {n:48s} {med:8.1f} clips/s  {statistics.median(m for _, m in vs):7.3f} ms/step  x{med / base:.4f}  runs {[v for v, _ in vs]}")
PY
rm -f "$out"
