#!/usr/bin/env python3
"""Calibration, not a product path: what the REFERENCE'S OWN ROUTE reaches on this GPU.  The reference is plain PyTorch -- nn.Linear /
nn.LayerNorm / nn.GELU / Conv3d modules under torch.cuda.amp.autocast, FlashAttention (or softmax(q k^T) v) for the attention core,
torch.optim.AdamW, a stack-of-norms gradient norm (modeling_finetune.py:37-166, engine_for_finetuning.py:64-100, utils.py:386-427) -- so the
same operator sequence written with stock torch modules (a generic pre-LN video ViT, nothing from the reference's files, nothing from this
library's kernels or its oracle) and run under bf16 autocast is the reference's arithmetic path on MI355X: hipBLASLt GEMMs, torch's fused
attention, torch's elementwise / LayerNorm kernels.  BASELINE configs[2]: ViT-B/16, 16 x 224 x 224, 32 clips, forward + CE + backward + AdamW.

    python tools/bench_torch_eager.py [--attn sdpa|naive] [--steps 10] [--warmup 3] [--batch 32]
"""
import argparse
import json
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

D, L, HEADS, FRAMES, NCLS = 768, 12, 12, 16, 2


def run(attn="sdpa", batch=32, steps=10, warmup=3, device=None):
    """one measurement; returns the result dict (bench.py's `torch_route` extra calls this too)"""

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.qkv = nn.Linear(D, 3 * D, bias=False)
            self.q_bias = nn.Parameter(torch.zeros(D))
            self.v_bias = nn.Parameter(torch.zeros(D))
            self.proj = nn.Linear(D, D)

        def forward(self, x):
            B, N, _ = x.shape
            bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias), self.v_bias))
            q, k, v = F.linear(x, self.qkv.weight, bias).reshape(B, N, 3, HEADS, D // HEADS).permute(2, 0, 3, 1, 4)
            if attn == "sdpa":
                o = F.scaled_dot_product_attention(q, k, v)
            else:
                o = ((q * (D // HEADS) ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1) @ v
            return self.proj(o.transpose(1, 2).reshape(B, N, D))

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm1, self.norm2 = nn.LayerNorm(D, eps=1e-6), nn.LayerNorm(D, eps=1e-6)
            self.attn = Attn()
            self.fc1, self.fc2 = nn.Linear(D, 4 * D), nn.Linear(4 * D, D)

        def forward(self, x):
            x = x + self.attn(self.norm1(x))
            return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))

    class VideoViT(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Conv3d(3, D, kernel_size=(2, 16, 16), stride=(2, 16, 16))
            n = (FRAMES // 2) * 14 * 14
            pos = torch.arange(n)[:, None] / torch.pow(10000.0, 2 * (torch.arange(D) // 2) / D)[None]
            self.register_buffer("pos", torch.where(torch.arange(D) % 2 == 0, pos.sin(), pos.cos())[None].float(), persistent=False)
            self.blocks = nn.ModuleList(Block() for _ in range(L))
            self.fc_norm = nn.LayerNorm(D, eps=1e-6)
            self.head = nn.Linear(D, NCLS)

        def forward(self, x):
            x = self.proj(x).flatten(2).transpose(1, 2)
            x = x + self.pos.type_as(x)
            for b in self.blocks:
                x = b(x)
            return self.head(self.fc_norm(x.mean(1)))

    dev = device if device is not None else torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VideoViT().to(dev).train()
    for m in model.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
    x = torch.randn(batch, 3, FRAMES, 224, 224, device=dev)
    y = torch.randint(0, NCLS, (batch,), device=dev)
    params = list(model.parameters())

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x), y)
        loss.backward()
        norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in params]), 2.0)  # the reference's grad-norm definition
        opt.step()
        opt.zero_grad()
        return loss, norm

    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    N = (FRAMES // 2) * 196
    gf = (3 * L * (24 * N * D * D + 4 * N * N * D) + 2 * (2 * N * 1536 * D) + 6 * D * NCLS) / 1e9
    res = {"what": "stock-PyTorch video ViT-B/16 under bf16 autocast (the reference's operator route) on this GPU", "attention": attn,
           "clips_per_s": round(batch / ms * 1e3, 1), "ms_per_step": round(ms, 2), "batch": batch, "steps": steps, "warmup": warmup,
           "frac_of_bf16_mfma_roofline": round(batch / ms * 1e3 * gf / 2516.6e3, 4),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1), "torch": torch.__version__}
    del model, opt, x, y, params
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--attn", default="sdpa", choices=["sdpa", "naive"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    print(json.dumps(run(a.attn, a.batch, a.steps, a.warmup)))
