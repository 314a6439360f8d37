"""Weight-gradient GEMMs on a second HIP stream (ops.set_dw_side_stream): A/B of the training step, alternating rounds on one box.

    python tools/exp_dw_stream.py [--rounds 3] [--steps 10] [--batch 32]

Variants: side stream off / on, each with the persistent and the per-tile-grid Linear schedule (a persistent launch counts on finding
every CU free; a dW workgroup still resident there delays that CU's whole tile list).  Also checks that every gradient after one
backward pass is bitwise the same with the side stream on.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", default="vit_base_patch16_224")
    a = ap.parse_args()
    import simple_tad_amd as T
    from simple_tad_amd import engine as E, kernels as K, ops as O, parallel as P
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = T.create_model(a.model, pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                           drop_path_rate=0.1, init_scale=0.001, use_flash_attn=True).to(dev).train()
    dp = P.DataParallel(model)
    opt = E.create_optimizer(dp, lr=1e-4, weight_decay=0.05, layer_decay=0.75)
    scaler = E.NativeScalerWithGradNormCount(dp)
    crit = torch.nn.CrossEntropyLoss()
    params = list(model.parameters())
    x = torch.randn(a.batch, 3, 16, 224, 224, device=dev)
    y = torch.randint(0, 2, (a.batch,), device=dev)
    dp.zero_grad()

    # ---- bitwise check of the gradients (drop-path off: the mask is drawn per forward)
    model.eval()  # no drop-path randomness; gradients still flow
    grads = {}
    for on in (False, True):
        O.set_dw_side_stream(on)
        dp.zero_grad()
        with torch.enable_grad():
            loss = crit(dp(x[:4]), y[:4])
        loss.backward()
        torch.cuda.synchronize()
        grads[on] = dp.flat_grad.clone()
    same = bool(torch.equal(grads[False], grads[True]))
    print("gradients bitwise equal with the side stream:", same, "side launches:", O._dw_side["launches"], flush=True)
    model.train()
    dp.zero_grad()

    def step():
        loss = crit(dp(x), y)
        scaler(loss, opt, parameters=params, update_grad=True)
        dp.zero_grad()

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    variants = [("main stream, persistent", False, 1), ("side stream, persistent", True, 1), ("main stream, per-tile grids", False, 0),
                ("side stream, per-tile grids", True, 0)]
    res = {v[0]: [] for v in variants}
    for r in range(a.rounds):
        for name, on, pers in variants:
            O.set_dw_side_stream(on)
            K.linear_tuning(persistent=pers)
            timed(3)
            res[name].append(round(timed(a.steps), 3))
        print("round", r, {k: v[-1] for k, v in res.items()}, flush=True)
    O.set_dw_side_stream(False)
    K.linear_tuning(persistent=1)
    out = {"bitwise_equal": same, "ms_per_step": res, "mean": {k: round(sum(v) / len(v), 3) for k, v in res.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
