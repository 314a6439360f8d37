"""Host-side cost of one training step (enqueue time without device sync) next to the synchronised wall time; cProfile top entries.
Usage: python tools/cpu_overhead.py [--profile]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simple_tad_amd as T
from simple_tad_amd import engine as E
from simple_tad_amd.parallel import DataParallel

ap = argparse.ArgumentParser()
ap.add_argument("--profile", action="store_true")
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                       drop_path_rate=0.1, init_scale=0.001, use_flash_attn=True).to(dev).train()
dp = DataParallel(model)
opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
scaler = E.NativeScalerWithGradNormCount(dp)
crit = torch.nn.CrossEntropyLoss()
params = list(model.parameters())
x = torch.randn(a.batch, 3, 16, 224, 224, device=dev)
y = torch.randint(0, 2, (a.batch,), device=dev)


def step():
    t0 = time.perf_counter()
    loss = crit(dp(x), y)
    t1 = time.perf_counter()
    scaler(loss, opt, parameters=params, update_grad=True)
    t2 = time.perf_counter()
    dp.zero_grad()
    return t1 - t0, t2 - t1


for _ in range(3):
    step()
torch.cuda.synchronize()
for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f = b = 0.0
    for _ in range(5):
        df, db = step()
        f += df
        b += db
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / 5:.1f} ms/step (forward {1e3 * f / 5:.1f}, backward+opt {1e3 * b / 5:.1f});  synchronised {1e3 * (t2 - t0) / 5:.1f} ms/step")
if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
