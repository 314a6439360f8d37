"""Precision modes at the real ViT-B/16 16x224x224 shape against golden G11 (the reference's fp64 forward + CE loss + backward), and
their step time at the benchmark's batch: the table DESIGN.md section 4 quotes.

    python tools/exp_precision.py [--modes fast,half,precise] [--scale 4096] [--steps 10]

For each mode: features / logits rel-L2, loss, and over all 152 gradient tensors the median / worst of (a) the stored 256-element
slice's relative error, (b) that slice's error relative to the tensor's RMS, (c) the sum-of-squares error; then clips/s of the
fwd + bwd + AdamW step at 32 clips.  "half" runs its backward on loss-scaled gradients (--scale) and removes the scale before
comparing, as engine.NativeScalerWithGradNormCount / the fused AdamW do.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fast,half")
    ap.add_argument("--scale", type=float, default=4096.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    import simple_tad_amd as T
    from simple_tad_amd import engine
    import test_real_size as RS
    g = np.load(os.path.join(ROOT, "tests", "golden", "g11_vitb_grads.npz"), allow_pickle=True)
    out = {}
    for mode in a.modes.split(","):
        m, x, y = RS.build_vitb()
        m = m.cuda().train()
        T.set_precision(mode)
        try:
            scale = a.scale if mode == "half" else 1.0
            feats = m.forward_features(x.cuda())
            logits = m.head(feats)
            loss = F.cross_entropy(logits, y.cuda())
            (loss * scale).backward()
            torch.cuda.synchronize()
            grads = {k: (p.grad / scale) for k, p in m.named_parameters()}
            rel = {k: RS.head_err(v, g, "grad." + k) for k, v in grads.items()}
            rms = {k: RS.head_err_scaled(v, g, "grad." + k) for k, v in grads.items()}
            sq = {k: RS.sq_err(v, g, "grad." + k) for k, v in grads.items()}
            gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
            finite = all(bool(torch.isfinite(v).all()) for v in grads.values())
            r = {"features": RS.rell2(feats, g["features"]), "logits": RS.rell2(logits, g["logits"]), "loss": loss.item(),
                 "loss_ref": float(g["loss"]), "grad_norm_rel": abs(gn - float(g["grad_norm"])) / float(g["grad_norm"]), "finite": finite}
            for name, d in (("slice_rel", rel), ("slice_over_rms", rms), ("sqsum", sq)):
                w = max(d, key=d.get)
                r[name] = {"median": float(np.median(list(d.values()))), "worst": d[w], "worst_key": w,
                           "n_over_1e-3": int(sum(v > 1e-3 for v in d.values()))}
            del m, grads
            torch.cuda.empty_cache()
            if not a.no_time:
                m2, _, _ = RS.build_vitb()
                m2 = m2.cuda().train()
                opt = engine.create_optimizer(m2, lr=1e-4, weight_decay=0.05, layer_decay=0.75)
                scaler = engine.NativeScalerWithGradNormCount(m2)
                xb = torch.randn(a.batch, 3, 16, 224, 224, device="cuda")
                yb = torch.randint(0, 2, (a.batch,), device="cuda")
                crit = torch.nn.CrossEntropyLoss()

                def step():
                    loss = crit(m2(xb), yb)
                    scaler(loss, opt, parameters=list(m2.parameters()), update_grad=True)
                    opt.zero_grad()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps
                r["ms_per_step"] = dt * 1e3
                r["clips_per_s"] = a.batch / dt
                r["loss_scale"] = scaler.state_dict()["scale"]
                r["skipped_steps"] = scaler.skipped_steps
                del m2, opt
                torch.cuda.empty_cache()
        finally:
            T.set_precision("fast")
        out[mode] = r
        print(mode, json.dumps(r), flush=True)
    return out


if __name__ == "__main__":
    main()
