#!/usr/bin/env python3
"""MAE pre-training step on one MI355X (SURVEY 8f-2 / BASELINE configs[4]): pretrain_videomae_large_patch16_224, decoder depth 12
(jobs/dapt/pretrain_capdata_large.sh:33-36), tube mask 0.75 (392 visible / 1176 masked tokens of 1568), synthetic clips resident in
HBM.  One step = mask -> reconstruction target -> encoder on visible tokens -> decoder -> MSE -> backward -> fused AdamW.
Prints one JSON line (clips/s, ms/step, algorithmic TFLOP/s).  Not the headline bench (bench.py); a measurement of the next row.

    python tools/bench_pretrain.py [--model pretrain_videomae_base_patch16_224] [--batch 32] [--steps 10] [--warmup 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple_tad_amd as T  # noqa: E402
import simple_tad_amd.modeling_pretrain  # noqa: E402,F401
from simple_tad_amd import engine as E, engine_pretrain as EP, ops  # noqa: E402
from simple_tad_amd.masking_generator import TubeMaskingGenerator  # noqa: E402
from simple_tad_amd.parallel import DataParallel  # noqa: E402


def block_flops(n, d):  # forward, per clip (24 N D^2 + 4 N^2 D)
    return 24.0 * n * d * d + 4.0 * n * n * d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="pretrain_videomae_large_patch16_224")
    ap.add_argument("--decoder-depth", type=int, default=12)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mask-ratio", type=float, default=0.75)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = T.create_model(a.model, pretrained=False, drop_path_rate=0.0, decoder_depth=a.decoder_depth).to(dev).train()
    dp = DataParallel(model)
    opt = E.create_optimizer(dp, lr=3e-4, weight_decay=0.05, betas=(0.9, 0.95))
    scaler = E.NativeScalerWithGradNormCount(dp)
    np.random.seed(0)
    gen = TubeMaskingGenerator((8, 14, 14), a.mask_ratio)
    x = torch.randn(a.batch, 3, 16, 224, 224, device=dev)
    masks = [torch.from_numpy(np.stack([gen() for _ in range(a.batch)])).to(dev).bool() for _ in range(a.steps + a.warmup)]
    n_mask = gen.total_masks
    params = list(model.parameters())

    def step(i):
        labels = EP.reconstruction_target(x, masks[i], 16, 2, True, n_mask)
        loss = ops.MseLossFn.apply(dp(x, masks[i], num_masked=n_mask), labels)
        dp.zero_grad()
        scaler(loss, opt, parameters=params)
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        last = step(a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    enc, dec = model.encoder, model.decoder
    n_vis = 1568 - n_mask
    f_fwd = (2.0 * 1568 * 1536 * enc.embed_dim + len(enc.blocks) * block_flops(n_vis, enc.embed_dim)
             + 2.0 * n_vis * enc.embed_dim * dec.embed_dim + len(dec.blocks) * block_flops(1568, dec.embed_dim)
             + 2.0 * n_mask * dec.embed_dim * 1536)
    f_step = 3.0 * f_fwd - 2.0 * 1568 * 1536 * enc.embed_dim  # patch-embed backward is dW only
    print(json.dumps({"metric": "clips/sec MAE pre-training step (fwd+bwd+AdamW)", "value": round(a.batch / dt, 2), "unit": "clips/sec",
                      "ms_per_step": round(1e3 * dt, 3), "model": a.model, "decoder_depth": a.decoder_depth, "batch": a.batch,
                      "visible_tokens": n_vis, "masked_tokens": n_mask, "algorithmic_gflop_per_clip": round(f_step / 1e9, 1),
                      "tflops": round(a.batch * f_step / dt / 1e12, 1), "frac_of_bf16_mfma_roofline": round(a.batch * f_step / dt / 2516.6e12, 4),
                      "loss": float(last), "params_m": round(sum(p.numel() for p in params) / 1e6, 1), "data": "synthetic"}))


if __name__ == "__main__":
    main()
