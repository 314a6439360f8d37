#!/bin/bash
# Run a list of GPU steps one after the other on the gpurun box: `tools/gpu_steps.sh OUTDIR "name|timeout_s|command" ...`
# Each step's output goes to OUTDIR/name.log.  An ordinary failure (non-zero exit) is recorded and the next step still runs; a step that
# is KILLED at its time limit (124 / 137) ends the run -- no further GPU step is started after a hang.
out="$1"; shift
mkdir -p "$out"
: > "$out/steps.txt"
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${tmo}s)"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -o pipefail -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "$name rc=$rc seconds=$(( $(date +%s) - start ))" | tee -a "$out/steps.txt"
  tail -n 3 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping" | tee -a "$out/steps.txt"; exit 1; fi
done
exit 0
