#!/usr/bin/env python3
"""Frame-by-frame inference latency (run_inference.py:84-101 flow): push one uint8 frame into the GPU ring buffer, predict on the
16-frame window, batch 1.  Reports ms per frame for the eager path and for HIP-graph replay.  python tools/bench_inference.py [--model ...]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple_tad_amd as T  # noqa: E402
from simple_tad_amd.inference import SlidingWindow  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_small_patch16_224")
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
torch.manual_seed(0)
m = T.create_model(a.model, pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm", init_scale=0.001).cuda().eval()
rng = np.random.RandomState(0)
frames = [rng.randint(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(32)]
out = {}
for mode in ("eager", "graph"):
    sw = SlidingWindow(m, bgr=True, use_graph=(mode == "graph"))
    for i in range(16 + 16):  # fill + one full lap of the ring (captures every offset in graph mode)
        sw.push(frames[i % 32])
        if sw.full:
            sw.predict()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        sw.push(frames[i % 32])
        r = sw.predict()
    r.cpu()
    out[mode + "_ms_per_frame"] = round(1e3 * (time.perf_counter() - t0) / a.frames, 3)
print(json.dumps({"model": a.model, **out, "fps_graph": round(1e3 / out["graph_ms_per_frame"], 1)}))
