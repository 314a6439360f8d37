#!/bin/bash
# The first multi-GPU run as ONE command (run ON an 8-GPU MI355X node from the repo root; nothing here was ever executed with more than
# one GPU -- no round had such a node -- so every step reports what it saw and the script keeps going past a failed configuration):
#
#   tools/scale_run.sh [out_dir] [max_gpus] [steps]          default: gpurun_out/scale  8  20
#
#   1. weak scaling   : bench.py --gpus N, 32 clips per GPU, N = 1, 2, 4, 8, each with the Linear GEMMs per-tile (DataParallel's default at
#                       world > 1) AND persistent (the single-GPU schedule): the choice between the two was made with two gloo ranks sharing
#                       one GPU and this run is what decides it
#   2. strong scaling : bench.py --gpus N --global-batch 256 (BASELINE configs[3]: 256 clips per step whatever N), both schedules
#   3. bf16 buckets   : the weak N = max run again with --bucket-dtype bf16 (172 instead of 345 MB on the xGMI ring per step)
#   4. one table      : clips/s, ms/step, speed-up over N = 1 of the same schedule, exposed exchange ms, RCCL version / channels / transport
#
# bench.py starts its own ranks (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1): one process per GPU,
# RCCL over xGMI.  Each run prints ONE JSON line; the lines are kept in $O/*.json, the table in $O/table.txt.
set -u
O="${1:-gpurun_out/scale}"; MAXN="${2:-8}"; STEPS="${3:-20}"; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
echo "[scale_run] visible GPUs: $have, requested up to $MAXN, $STEPS timed steps per run" | tee "$O/table.txt"
run() {  # name, gpus, extra flags...
  local name="$1" n="$2"; shift 2
  if [ "$n" -gt "$have" ]; then echo "[scale_run] skip $name: needs $n GPUs, $have visible" | tee -a "$O/skipped.txt"; return 0; fi
  echo "[scale_run] $name: python3 bench.py --gpus $n --steps $STEPS --warmup 5 --no-extras --no-cpu-baseline $*"
  timeout -k 10 900 python3 bench.py --gpus "$n" --steps "$STEPS" --warmup 5 --no-extras --no-cpu-baseline "$@" > "$O/$name.out" 2> "$O/$name.err"
  local rc=$?
  grep '^{' "$O/$name.out" | tail -1 > "$O/$name.json"
  if [ $rc -ne 0 ] || [ ! -s "$O/$name.json" ]; then echo "[scale_run] $name FAILED (rc $rc):"; tail -5 "$O/$name.err"; rm -f "$O/$name.json"; fi
  return 0
}
for n in 1 2 4 8; do
  [ "$n" -gt "$MAXN" ] && continue
  for sched in per-tile persistent; do
    # (one rank: DataParallel sets no schedule -- the single-GPU plan IS persistent -- so the N = 1 row is measured once per table)
    if [ "$n" -eq 1 ]; then [ "$sched" = persistent ] && run "weak_n1_persistent" 1; continue; fi
    run "weak_n${n}_${sched}" "$n" --linear-schedule "$sched"
    run "strong_n${n}_${sched}" "$n" --global-batch 256 --linear-schedule "$sched"
  done
done
[ "$have" -ge 2 ] && run "weak_n$(( have < MAXN ? have : MAXN ))_per-tile_bf16" "$(( have < MAXN ? have : MAXN ))" --linear-schedule per-tile --bucket-dtype bf16
run "strong_n1_persistent" 1 --global-batch 256
python3 - "$O" <<'PY' | tee -a "$O/table.txt"
import glob, json, os, sys
O = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(O, "*.json"))):
    try:
        rows[os.path.basename(f)[:-5]] = json.loads(open(f).read())
    except Exception as e:  # noqa: BLE001
        print(f"[scale_run] unreadable {f}: {e!r}")
base = {k.split("_")[0]: v for k, v in rows.items() if "_n1_" in k}
print(f"{'run':34s} {'N':>2s} {'clips/GPU':>9s} {'clips/s':>10s} {'ms/step':>9s} {'x N=1':>7s} {'exposed ms':>10s} {'buckets':>7s} {'MB/step':>8s}  rank ms/step min..max")
for k, v in sorted(rows.items(), key=lambda kv: (kv[0].split("_")[0], kv[1]["n_gpus"], kv[0])):
    c = v.get("collective") or {}
    b = base.get(k.split("_")[0])
    sp = f"{v['value'] / b['value']:.2f}" if b else "-"
    rk = c.get("rank_ms_per_step") or {}
    print(f"{k:34s} {v['n_gpus']:2d} {v['config']['per_gpu_batch']:9d} {v['value']:10.1f} {v['ms_per_step']:9.2f} {sp:>7s} "
          f"{str(c.get('exposed_ms', '-')):>10s} {str(c.get('buckets', '-')):>7s} {c.get('allreduce_bytes_per_step', 0) / 1e6:8.1f}  "
          f"{rk.get('min', '-')}..{rk.get('max', '-')}")
for k, v in rows.items():
    r = (v.get("collective") or {}).get("rccl")
    if r:
        print(f"[scale_run] RCCL (from {k}): version {r.get('version')}, channels {r.get('channels')}, transports {r.get('transports')}, "
              f"rank 0 -> peers {r.get('peers_of_rank0')}: {r.get('transport_of_rank0')}")
        for ln in r.get("lines", [])[:6]:
            print("            ", ln)
        break
print("[scale_run] weak rows: x N=1 is the speed-up of the whole job (ideal = N); strong rows: 256 clips per step whatever N (ideal = N as well). "
      "`exposed ms` = how long backward's tail waited for the gradient exchange (DataParallel.timing_summary).")
PY
