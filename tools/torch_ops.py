#!/usr/bin/env python3
"""Which PyTorch (non-tad) device kernels does one training step launch, and from where?  torch.profiler over one step of the
bench workload; prints aten ops that launched kernels with input shapes and the innermost repo stack frame."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simple_tad_amd as T
from simple_tad_amd import engine as E
from simple_tad_amd.parallel import DataParallel

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                       drop_path_rate=0.1, init_scale=0.001, use_flash_attn=True).to(dev)
x = torch.randn(B, 3, 16, 224, 224, device=dev)
y = torch.randint(0, 2, (B,), device=dev)
model.train()
dp = DataParallel(model, bucket_mb=64.0)
opt = E.create_optimizer(dp, lr=1e-3, weight_decay=0.05, layer_decay=0.75)
scaler = E.NativeScalerWithGradNormCount(dp)
crit = torch.nn.CrossEntropyLoss()
params = list(model.parameters())
dp.zero_grad()

def step():
    loss = crit(dp(x), y)
    scaler(loss, opt, parameters=params, update_grad=True)
    dp.zero_grad()
    return loss

for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if not any(k for k in ev.kernels) and ev.name not in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::zero_", "aten::fill_"):
        continue
    frame = next((f for f in ev.stack if "simple_tad_amd/" in f or "bench" in f or "torch_ops" in f), ev.stack[0] if ev.stack else "?")
    agg[(ev.name, str(ev.input_shapes)[:60], frame.split("/root/repo/")[-1][:90] if "/root/repo/" in frame else frame[-90:])] += 1
mem = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == "CPU" and any("emcpy" in k.name or "emset" in k.name for k in ev.kernels):
        frame = next((f for f in ev.stack if "simple_tad_amd/" in f or "bench" in f or "torch_ops" in f), ev.stack[0] if ev.stack else "?")
        mem[(ev.name, str(ev.input_shapes)[:60], [k.name for k in ev.kernels][0][:30], frame[-80:])] += 1
print("--- ops that issued a memcpy / memset")
for k, n in sorted(mem.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {k}")
print("--- aten ops")
for (name, shp, fr), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {name:28s} {shp:60s} {fr}")
