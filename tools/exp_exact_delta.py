"""A/B of the attention backward's exact delta (forward writes the rounding residual of its output, backward reads it) on the
benchmark step: alternating rounds in ONE process (ViT-B/16 16x224x224, 32 clips, fwd + bwd + AdamW)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple_tad_amd as T
from simple_tad_amd import engine as E, ops

torch.manual_seed(0)
m = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                   drop_path_rate=0.1, init_scale=0.001, use_flash_attn=True).cuda().train()
opt = E.create_optimizer(m, lr=1e-4, weight_decay=0.05, layer_decay=0.75)
sc = E.NativeScalerWithGradNormCount(m)
x = torch.randn(32, 3, 16, 224, 224, device="cuda"); y = torch.randint(0, 2, (32,), device="cuda")
crit = torch.nn.CrossEntropyLoss(); params = list(m.parameters())
def step():
    sc(crit(m(x), y), opt, parameters=params); opt.zero_grad()
def timed(n=10):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
res = {True: [], False: []}
for r in range(4):
    for on in (True, False):
        ops.set_attn_exact_delta(on); res[on].append(timed())
print("exact delta on :", ["%.3f" % v for v in res[True]], "ms/step")
print("exact delta off:", ["%.3f" % v for v in res[False]], "ms/step")
print("cost: %.3f ms/step" % (sum(res[True]) / 4 - sum(res[False]) / 4))
