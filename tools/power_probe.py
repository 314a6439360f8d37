#!/usr/bin/env python3
"""Board power and clocks while the hot path runs (run ON the GPU box): is the step bound by the chip's power limit?

    python tools/power_probe.py [--seconds 4] [--only step,gemm_nt,...] [--out gpurun_out/x/power.json]

A sampler thread reads the amdgpu hwmon files of the device (power1_average / power1_input, power1_cap, freq1_input = shader clock) every
20 ms -- or amdsmi's socket power when sysfs is not readable -- while the main thread runs ONE kernel class back to back on random data
for `--seconds` (the first second is dropped: the power controller needs it to settle).  Per class: mean / max power, the cap, the mean
shader clock, and the rate the loop ran at.  Energy per step of a class = its mean power x its ms/step (bench.py's roofline_all).
What the table says (MI355X_MICROARCH.md, DVFS give-back): a class that draws the cap is bound by ENERGY -- cycles saved in it come back
as a lower clock unless they also save energy (fewer bytes from beyond L2, fewer VALU / LDS operations per MFMA) -- and a class well under
the cap is bound by something else (HBM, latency, issue).
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler:
    def __init__(self, period=0.02):
        self.period, self.samples, self._stop, self._th = period, [], threading.Event(), None
        self.src, self.cap_w = None, None
        hw = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            p = next((os.path.join(d, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))), None)
            if p:
                hw.append((d, p))
        # the visible device is the one HIP uses; with several cards in sysfs take the one whose power moves (decided in pick())
        self._hw = hw
        self._smi = None
        if not hw:
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                self._smi = (amdsmi, amdsmi.amdsmi_get_processor_handles())
            except Exception as e:  # noqa: BLE001
                self.src = f"no power source: {e!r}"

    def _read_all(self):
        out = []
        for d, p in self._hw:
            try:
                w = int(open(p).read()) / 1e6
                f = os.path.join(d, "freq1_input")
                mhz = int(open(f).read()) / 1e6 if os.path.exists(f) else None
                out.append((w, mhz))
            except Exception:  # noqa: BLE001
                out.append((None, None))
        if self._smi:
            amdsmi, hs = self._smi
            for h in hs:
                try:
                    pi = amdsmi.amdsmi_get_power_info(h)
                    w = pi.get("current_socket_power") or pi.get("average_socket_power")
                    out.append((float(w) if w not in (None, "N/A") else None, None))
                except Exception:  # noqa: BLE001
                    out.append((None, None))
        return out

    def caps(self):
        caps = []
        for d, _ in self._hw:
            try:
                caps.append(int(open(os.path.join(d, "power1_cap")).read()) / 1e6)
            except Exception:  # noqa: BLE001
                caps.append(None)
        if self._smi:
            amdsmi, hs = self._smi
            for h in hs:
                try:
                    caps.append(float(amdsmi.amdsmi_get_power_cap_info(h)["power_cap"]) / 1e6)
                except Exception:  # noqa: BLE001
                    caps.append(None)
        return caps

    def start(self):
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(), self._read_all()))
                time.sleep(self.period)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self, drop_s=1.0):
        self._stop.set()
        self._th.join()
        if not self.samples:
            return None
        t0 = self.samples[0][0]
        keep = [s for t, s in self.samples if t - t0 >= drop_s] or [s for _, s in self.samples]
        ncard = len(keep[0])
        per = []
        for c in range(ncard):
            ws = [s[c][0] for s in keep if s[c][0] is not None]
            fs = [s[c][1] for s in keep if s[c][1] is not None]
            per.append({"mean_w": sum(ws) / len(ws) if ws else None, "max_w": max(ws) if ws else None, "mean_sclk_mhz": sum(fs) / len(fs) if fs else None, "n": len(ws)})
        return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--cmd", default="", help="sample while this command runs (a child process, e.g. tools/micro/mfma_power 16 5) instead of the built-in classes")
    a = ap.parse_args()
    if a.cmd:
        import shlex
        import subprocess
        sm = Sampler()
        caps = sm.caps()
        sm.start()
        t0 = time.perf_counter()
        pr = subprocess.run(shlex.split(a.cmd), capture_output=True, text=True)
        dt = time.perf_counter() - t0
        per = sm.stop(drop_s=1.5)
        print(pr.stdout[-1500:], pr.stderr[-300:])
        card = max(range(len(per)), key=lambda i: per[i]["mean_w"] or 0.0) if per else None
        res = {"cmd": a.cmd, "seconds": dt, "caps_w": caps, "cards": per, "loaded_card_index": card, "stdout_tail": pr.stdout.strip().splitlines()[-3:]}
        if card is not None:
            c = per[card]
            print(f"[power_probe] {a.cmd}: mean {c['mean_w']:.0f} W, max {c['max_w']:.0f} W of a {caps[card]} W cap, mean shader clock {c['mean_sclk_mhz'] or 0:.0f} MHz "
                  f"({c['n']} samples after the first 1.5 s)")
        if a.out:
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            json.dump(res, open(a.out, "w"), indent=1)
        return
    import simple_tad_amd as T
    from simple_tad_amd import engine as E, kernels as K
    dev, bf = torch.device("cuda", 0), torch.bfloat16
    B, N, D, H = 32, 1568, 768, 12
    M = B * N

    def rnd(*shape, dtype=bf, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    x, w3, w4 = rnd(M, D), rnd(3 * D, D, scale=0.02), rnd(4 * D, D, scale=0.02)
    x4, w1 = rnd(M, 4 * D), rnd(D, 4 * D, scale=0.02)
    b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
    res = torch.randn(M, D, device=dev)
    dy4 = rnd(M, 4 * D)
    qkv = rnd(M, 3 * D)
    qkv[:, :D] = (qkv[:, :D].float() * K.q_prescale_of(0.125)).to(bf)
    out, lse, lo = K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True, q_prescaled=True)
    dout = rnd(M, D)
    xf = torch.randn(M, D, device=dev)
    g, bb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    _, mean, rstd = K.layernorm_fwd(xf, g, bb, 1e-6)

    only = [o for o in a.only.split(",") if o]
    need_model = not only or any(n.startswith(o) for o in only for n in ("step", "forward_only"))
    step = fwd_only = None
    if need_model:
      torch.manual_seed(0)
      model = T.create_model(  "vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                             drop_path_rate=0.1, init_scale=0.001, use_flash_attn=True).to(dev).train()
      from simple_tad_amd.parallel import DataParallel
      dp = DataParallel(model, bucket_mb=64.0)
      opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
      scaler = E.NativeScalerWithGradNormCount(dp)
      crit = torch.nn.CrossEntropyLoss()
      clip = torch.randn(B, 3, 16, 224, 224, device=dev)
      y = torch.randint(0, 2, (B,), device=dev)
      params = list(model.parameters())
      dp.zero_grad()

      def step():
          scaler(crit(dp(clip), y), opt, parameters=params, update_grad=True)
          dp.zero_grad()

      def fwd_only():
          with torch.no_grad():
              model(clip)

    work = {
        "idle": (None, 0.0),
        "step": (step, B * 1078.37e9),
        "forward_only": (fwd_only, B * 360.69e9),
        "gemm_nt qkv (K=768, bias)": (lambda: K.linear_fwd(x, w3, b3), 2.0 * M * 3 * D * D),
        "gemm_nt fc1 (K=768, GELU)": (lambda: K.linear_fwd(x, w4, b4, epilogue=1, want_preact=True), 2.0 * M * 4 * D * D),
        "gemm_nt fc2 (K=3072, +residual f32)": (lambda: K.linear_fwd(x4, w1, b1, out_dtype=torch.float32, epilogue=2, residual=res), 2.0 * M * 4 * D * D),
        "gemm_tn dW fc1": (lambda: K.linear_bwd_weight(dy4, x, want_bias=True), 2.0 * M * 4 * D * D),
        "attn_fwd": (lambda: K.attn_fwd(qkv, B, N, H, 0.125, q_prescaled=True), 4.0 * B * H * N * N * 64),
        "attn_bwd": (lambda: K.attn_bwd(qkv, out, dout, lse, B, N, H, 0.125, out_lo=lo, q_prescaled=True), 8.0 * B * H * N * N * 64),
        "layernorm_bwd": (lambda: K.layernorm_bwd(dout, xf, g, mean, rstd, dres=xf, want_bf16=True, want_colsum=True), 0.0),
    }
    sm = Sampler()
    caps = sm.caps()
    rows = {}
    for name, (fn, flops) in work.items():
        if (only and not any(name.startswith(o) for o in only)) or (fn is None and name != "idle"):
            continue
        if fn is not None:
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        sm.start()
        t0 = time.perf_counter()
        n = 0
        if fn is None:
            time.sleep(a.seconds)
        else:
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
                n += 8
        dt = time.perf_counter() - t0
        per = sm.stop()
        rows[name] = {"calls": n, "ms_per_call": 1e3 * dt / n if n else None, "tflops": flops * n / dt / 1e12 if n and flops else None, "cards": per}
        print(name, json.dumps(rows[name]), flush=True)
    # the card that carries the load is the one whose power moves most between idle and the step
    res_out = {"caps_w": caps, "source": "sysfs hwmon" if sm._hw else ("amdsmi" if sm._smi else sm.src), "classes": rows, "device": K.device_info()}
    loaded = [r for n, r in rows.items() if n != "idle" and r["cards"]]
    if loaded:
        ncard = len(loaded[0]["cards"])
        card = max(range(ncard), key=lambda i: max((r["cards"][i]["mean_w"] or 0) for r in loaded))
        res_out["loaded_card_index"] = card
        print(f"\n{'class':40s} {'W mean':>8s} {'W max':>8s} {'cap':>6s} {'% cap':>6s} {'sclk MHz':>9s} {'ms/call':>9s} {'TFLOP/s':>8s} {'J/call':>8s}")
        for name, r in rows.items():
            c = r["cards"][card]
            cap = caps[card] if card < len(caps) else None
            j = (c["mean_w"] * r["ms_per_call"] * 1e-3) if (c["mean_w"] and r["ms_per_call"]) else None
            print(f"{name:40s} {c['mean_w'] or 0:8.0f} {c['max_w'] or 0:8.0f} {cap or 0:6.0f} {100 * (c['mean_w'] or 0) / cap if cap else 0:6.1f} "
                  f"{c['mean_sclk_mhz'] or 0:9.0f} {r['ms_per_call'] or 0:9.3f} {r['tflops'] or 0:8.0f} {j or 0:8.2f}")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res_out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
