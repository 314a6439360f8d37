#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the ViT-B/16 16x224x224 fine-tuning step (forward + backward + gradient
all-reduce + AdamW) on N MI355X GPUs of one node -- BASELINE.json configs[2] (N=1) / configs[3] (N=8).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch (32 clips per GPU, resident in HBM):
per-step lr assignment, forward, CE loss, backward (bucketed RCCL all-reduce overlapped), grad-norm,
AdamW step, zero_grad.  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around
launches of the dominant kernel (gemm_nt_kernel, the bf16 MFMA GEMM) inside the timed region -- every 13th
launch (13 is coprime with the 96 launches per step, so all GEMM shapes are sampled equally; a timed event
pair costs ~20 us of queue time, and bracketing all 96 launches per step slowed the step by 3.6 %);
`cpu_baseline` times the oracle (CPU restatement of the reference path) on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBS = 8000.0


def flops_per_clip(N, D, L, n_cls=2, k_patch=1536):
    """algorithmic FLOPs (2/MAC, full N^2 attention, no recompute, patch-embed bwd = dW only)."""
    f_patch = 2 * N * k_patch * D
    f_blk = 24 * N * D * D + 4 * N * N * D
    return f_patch + L * f_blk + 2 * D * n_cls, 3 * L * f_blk + 2 * f_patch + 6 * D * n_cls


def cpu_baseline(state_dict, frames, reps=2):
    """Oracle (pure-torch CPU restatement of the reference path) fwd+bwd, B=2, fp32, all host cores."""
    from oracle import vit_oracle as O
    # a 256-thread pool on this small batch is slower than 32 threads (oversubscription): use at most 32 cores
    torch.set_num_threads(min(os.cpu_count(), 32))
    P = {k: v.detach().float().cpu().requires_grad_() for k, v in state_dict.items()}
    torch.manual_seed(0)
    x = torch.randn(2, 3, frames, 224, 224)
    y = torch.randint(0, 2, (2,))

    def one():
        for p in P.values():
            p.grad = None
        logits = O.forward(x, P, depth=12, num_heads=12, tubelet=2, patch=16)
        torch.nn.functional.cross_entropy(logits, y).backward()

    t0 = time.perf_counter()
    one()  # warm-up
    warm = time.perf_counter() - t0
    if warm > 20.0:  # keep the default bench run bounded: a slow host reports the warm-up pass itself
        reps, dt = 0, warm
    else:
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        dt = (time.perf_counter() - t0) / reps
    return {"value": round(2 / dt, 4), "unit": "clips/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"ViT-B/16 16x224x224 fwd+bwd (CE loss), batch 2, fp32, {reps} reps after 1 warm-up, oracle/vit_oracle.py "
                      f"using {torch.get_num_threads()} of {os.cpu_count()} host cores; {dt:.2f} s per batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--drop-path", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-profile", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-class table to stderr")
    ap.add_argument("--graph", type=int, default=-1, help="1: capture the whole step in a HIP graph and replay it (N=1 only); default eager")
    args = ap.parse_args()

    import simple_tad_amd as T
    from simple_tad_amd import engine as E
    from simple_tad_amd import kernels as K
    from simple_tad_amd import _lib
    from simple_tad_amd.parallel import DataParallel, init_distributed_mode

    distributed, rank, world, local = init_distributed_mode()
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1", file=sys.stderr)
        if args.gpus > 1 and world == 1:
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the MI355X path has no CPU fallback)"
    if os.environ.get("TAD_DIST_BACKEND") == "gloo":  # debugging aid: several ranks share the visible GPUs
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    info = K.device_info()

    torch.manual_seed(0)  # identical init on every rank
    model = T.create_model(args.model, pretrained=False, num_classes=2, all_frames=args.frames, tubelet_size=2,
                           final_reduction="fc_norm", drop_path_rate=args.drop_path if args.mode == "train" else 0.0,
                           init_scale=0.001, use_flash_attn=True).to(dev)
    sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()} if rank == 0 else None
    D, L = model.embed_dim, model.get_num_layers()
    ntok = model.patch_embed.num_patches
    f_fwd, f_fb = flops_per_clip(ntok, D, L)

    torch.manual_seed(0 + rank)  # per-rank data seed (run_class_finetuning.py:222)
    B = args.batch
    x = torch.randn(B, 3, args.frames, 224, 224, device=dev)
    y = torch.randint(0, 2, (B,), device=dev)
    total_steps = args.steps + args.warmup
    lr_sched = E.cosine_scheduler(5e-4 * B * world / 256, 1e-6, 1, max(total_steps, 2), warmup_epochs=0)

    if args.mode == "train":
        model.train()
        dp = DataParallel(model, bucket_mb=64.0)
        opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
        scaler = E.NativeScalerWithGradNormCount(dp)
        crit = torch.nn.CrossEntropyLoss()
        params = [p for p in model.parameters()]
        dp.zero_grad()

        def step(it):
            for g in opt.param_groups:
                g["lr"] = lr_sched[it] * g["lr_scale"]
            loss = crit(dp(x), y)
            scaler(loss, opt, parameters=params, update_grad=True)
            dp.zero_grad()
            return loss

        use_graph = (args.graph == 1)  # measured: the step is not launch-bound (eager 61.8 ms vs graph 62.3 ms), so eager is the default
        if use_graph and world == 1:
            # Launch-bound stretches (hundreds of short launches per step) are removed by capturing the whole step -- forward,
            # loss, backward, grad-norm, AdamW, zero_grad -- into ONE HIP graph and replaying it.  The learning rate lives in a
            # device tensor per param group (capturable AdamW) that is refreshed before each replay, so the schedule still applies.
            opt = torch.optim.AdamW([{"params": g["params"], "weight_decay": g["weight_decay"], "lr_scale": g["lr_scale"],
                                      "lr": torch.tensor(float(g["lr"]), device=dev)} for g in opt.param_groups],
                                    betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=True, fused=True)
            lr_table = torch.tensor(lr_sched, dtype=torch.float32, device=dev)
            scales = [g["lr_scale"] for g in opt.param_groups]

            def eager_body():
                loss = crit(dp(x), y)
                scaler(loss, opt, parameters=params, update_grad=True)
                dp.zero_grad()
                return loss

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = eager_body()

            def step(it):  # noqa: F811
                for g, sc in zip(opt.param_groups, scales):
                    g["lr"].copy_(lr_table[it] * sc)
                graph.replay()
                return static_loss
    else:
        model.eval()

        def step(it):
            with torch.no_grad():
                return model(x)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step(it)
    prof = None
    if rank == 0 and not args.no_live_profile:
        prof = K.LaunchProfiler(only=None if args.breakdown else ["gemm_nt"], stride=1 if args.breakdown else 13)
        K.set_profiler(prof)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        last = step(args.warmup + it)
    barrier()
    dt = time.perf_counter() - t0
    K.set_profiler(None)
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(last.float().mean().item()) if args.mode == "train" else float("nan")

    if rank != 0:
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    clips_per_s = B * world * args.steps / dt
    fl = f_fb if args.mode == "train" else f_fwd
    out = {
        "metric": "clips/sec (16x224^2 ViT-B) " + ("fwd+bwd" if args.mode == "train" else "forward"),
        "value": round(clips_per_s, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} {args.frames}x224x224, {B} clips/GPU, " +
                               ("fwd+bwd+AdamW fine-tune step with CE loss on synthetic labels (BASELINE configs[2]/[3])"
                                if args.mode == "train" else "forward only (BASELINE configs[1])"),
                   "global_batch": B * world, "per_gpu_batch": B, "tokens_per_clip": ntok, "parallelism": f"dp{world}",
                   "drop_path": args.drop_path if args.mode == "train" else 0.0, "residual_stream": "f32", "operands": "bf16",
                   "algorithmic_gflop_per_clip": round(fl / 1e9, 2)},
        "frac_of_bf16_mfma_roofline": round(clips_per_s * fl / world / (PEAK_BF16_TFLOPS * 1e12), 4),
        "loss": loss_val,
        "device": info,
    }
    if prof is not None:
        summ = prof.summary()
        g = summ.get("gemm_nt")
        if g and g["ms"] > 0:
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            traffic = None  # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
            try:
                import glob
                latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[-1]
                traffic = round(json.load(open(latest)).get("gemm_nt_traffic_bytes_per_launch"))
            except Exception:  # noqa: BLE001
                traffic = None
            out["roofline"] = {"kernel": "gemm_nt_kernel (bf16 MFMA GEMM, all Linear fwd / input-grad launches)", "bound": "mfma",
                               "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                               # a Linear call is one kernel launch, or two under the split-tail plan (whole rounds of 256x256
                               # tiles + remaining rows): times and flops are per KERNEL launch, as rocprofv3 counts them
                               "calls": prof.seen.get("gemm_nt", g["calls"]), "sampled_calls": g["calls"], "sampled_launches": g["launches"],
                               "launches": round(prof.seen.get("gemm_nt", g["calls"]) * g["launches"] / max(g["calls"], 1)),
                               "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
                               "gflop_per_launch": round(g["flops"] / g["launches"] / 1e9, 2)}
        tot = sum(v["ms"] for v in summ.values())
        if args.breakdown:
            out["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            print(f"[bench] per-kernel-class (events, {args.steps} steps):", file=sys.stderr)
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
                print(f"   {k:16s} {v['launches']:6d} launches {v['ms'] / args.steps:9.3f} ms/step  {tf:8.1f} TFLOP/s  {gbs:8.1f} GB/s(alg)",
                      file=sys.stderr)
            print(f"   sum of launches {tot / args.steps:.3f} ms/step vs wall {1e3 * dt / args.steps:.3f} ms/step", file=sys.stderr)
    if world == 1 and not args.no_cpu_baseline and args.model == "vit_base_patch16_224":
        try:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, args.frames)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
