"""Deterministic input/weight recipes shared by ``tools/make_goldens.py`` (which
feeds them to the imported reference) and by the tests (which feed the same
tensors to the oracle and to the HIP path).  Keeping the recipe here means the
committed fixtures only need to hold the reference's *outputs*.

Everything is generated on CPU with an explicit ``torch.Generator`` so the
stream does not depend on global RNG state.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _seed_for(tag: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{tag}".encode()).digest()
    return int.from_bytes(h[:4], "little")


def tensor_for(tag: str, shape: Iterable[int], seed: int = 0, scale: float = 1.0, shift: float = 0.0,
               dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(tag, seed))
    return (torch.randn(*shape, generator=g, dtype=torch.float32) * scale + shift).to(dtype)


def uint8_for(tag: str, shape: Iterable[int], seed: int = 0) -> torch.Tensor:
    """seeded uint8 image data (every byte value occurs)"""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(tag, seed))
    return torch.randint(0, 256, tuple(shape), generator=g, dtype=torch.uint8)


def eval_probs_labels(tag: str, n: int, seed: int = 0):
    """class-1 probabilities (float32 numpy, informative but noisy; a third of them rounded to the 0.01 grid so that scores tie
    with each other and with the thresholds) and binary labels"""
    import numpy as np
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(tag, seed))
    labels = (torch.rand(n, generator=g) < 0.3).long()
    logit = torch.randn(n, generator=g) * 1.5 + (labels.float() * 2.0 - 1.0)
    probs = torch.sigmoid(logit)
    grid = torch.rand(n, generator=g) < 0.33
    probs = torch.where(grid, (probs * 100).round() / 100, probs).float()
    return probs.numpy().astype(np.float32), labels.numpy()


def params_for(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Weights: N(0, 0.02) for matrices/conv kernels, N(0, 0.02) biases (so that
    biases are exercised), N(1, 0.02) for LayerNorm / layer-scale weights."""
    out = {}
    for name, shape in shapes.items():
        is_norm_w = (name.endswith("norm1.weight") or name.endswith("norm2.weight")
                     or name.endswith("fc_norm.weight") or name == "norm.weight")
        out[name] = tensor_for(name, shape, seed, scale=0.02, shift=1.0 if is_norm_w else 0.0)
    return out


def vit_param_shapes(embed_dim: int, depth: int, num_classes: int, in_chans: int = 3, tubelet: int = 2,
                     patch: int = 16, mlp_ratio: int = 4, final_reduction: str = "fc_norm"):
    """State-dict keys/shapes of the reference VisionTransformer with qkv_bias=True
    (SURVEY.md section 8b), in the reference's registration order."""
    D = embed_dim
    s = {"patch_embed.proj.weight": (D, in_chans, tubelet, patch, patch), "patch_embed.proj.bias": (D,)}
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.q_bias"] = (D,)
        s[p + "attn.v_bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (mlp_ratio * D, D)
        s[p + "mlp.fc1.bias"] = (mlp_ratio * D,)
        s[p + "mlp.fc2.weight"] = (D, mlp_ratio * D)
        s[p + "mlp.fc2.bias"] = (D,)
    if final_reduction == "fc_norm":
        s["fc_norm.weight"] = (D,)
        s["fc_norm.bias"] = (D,)
    else:
        s["norm.weight"] = (D,)
        s["norm.bias"] = (D,)
    s["head.weight"] = (num_classes, D)
    s["head.bias"] = (num_classes,)
    return s


def summarize(t: torch.Tensor, head: int = 256) -> Dict[str, np.ndarray]:
    """Compact pin for a large tensor: fp64 sum, fp64 sum of squares and the
    first ``head`` flattened elements."""
    f = t.detach().double().flatten()
    return {
        "sum": np.array(f.sum().item()),
        "sqsum": np.array((f * f).sum().item()),
        "head": t.detach().flatten()[:head].clone().numpy(),
    }


def check_summary(t: torch.Tensor, npz, key: str, rtol: float, atol_scale: float = 1.0):
    """Assert tensor ``t`` matches the stored summary ``key.*`` in ``npz``."""
    f = t.detach().double().flatten()
    head = torch.from_numpy(npz[key + ".head"]).double()
    n = head.numel()
    scale = max(head.abs().max().item(), 1e-30)
    err = (f[:n] - head).abs().max().item() / scale
    assert err <= rtol, f"{key}: head mismatch rel {err:.3e} > {rtol}"
    sq = float(npz[key + ".sqsum"])
    got = (f * f).sum().item()
    assert abs(got - sq) <= 4 * rtol * max(sq, 1e-30) * atol_scale, f"{key}: sqsum {got} vs {sq}"


# the tiny full-model configuration used by G3 (real head geometry d=64, K_patch % 64 == 0)
TINY = dict(img_size=16, patch_size=8, embed_dim=128, depth=2, num_heads=2, all_frames=4, tubelet_size=2,
            num_classes=2)


# ---- G12: fine-tune trajectory of the tiny model through the reference's real engine (three optimizer steps, update_freq 2)
G12 = dict(micro_batches=6, update_freq=2, steps=3, base_lr=2e-3, min_lr=1e-5, warmup_epochs=1, warmup_steps=1, start_warmup_value=2e-4, weight_decay=0.05, weight_decay_end=0.1,
           layer_decay=0.75, clip_grad=1.5, labels=[[0, 1], [1, 1], [0, 0], [1, 0], [0, 1], [1, 1]])


def g12_batches(dtype=torch.float32):
    """the six micro-batches of G12: (samples [2,3,T,H,W], targets [2], None, None) -- the tuple engine_for_finetuning iterates over"""
    c = TINY
    return [(tensor_for(f"g12.x{i}", (2, 3, c["all_frames"], c["img_size"], c["img_size"]), seed=12 + i).to(dtype),
             torch.tensor(G12["labels"][i]), None, None) for i in range(G12["micro_batches"])]


# ---- G13: three MAE pre-training steps of the tiny model through the reference's real engine (AdamW, lr / wd schedules, clipping)
G13 = dict(steps=3, base_lr=1.5e-3, min_lr=1e-5, warmup_epochs=1, warmup_steps=1, start_warmup_value=3e-4, weight_decay=0.05,
           weight_decay_end=0.08, clip_grad=0.005, betas=(0.9, 0.95))


def g13_batches(dtype=torch.float32):
    """(videos [2,3,16,32,32], tube mask [2,32] bool) per step: masks from a seeded RandomState as in G8 (2x2 grid, 0.75, 8 slots)"""
    import numpy as np
    rng = np.random.RandomState(13)
    per = np.hstack([np.zeros(1), np.ones(3)])
    out = []
    for i in range(G13["steps"]):
        masks = []
        for _ in range(2):
            rng.shuffle(per)
            masks.append(np.tile(per, (8, 1)).flatten())
        out.append((tensor_for(f"g13.x{i}", (2, 3, 16, 32, 32), seed=130 + i).to(dtype), torch.from_numpy(np.stack(masks)).bool()))
    return out


# ---- real-size fixtures (G10 ViT-L/16 MAE, G11 ViT-B/16 gradients): weights come from the model's own seeded init, inputs from here
G10_ROWS = [0, 1, 587, 1175]   # masked-token rows of each clip whose whole 1536-wide prediction / label is stored


def rerandomize_1d(model, seed: int = 1234):
    """Every 1-D parameter ~ N(0, 0.02) (+1 for norm weights), in registration order, so that biases and LayerNorm affine terms
    are exercised (the seeded init leaves them at 0 / 1).  Same recipe as G4's inline loop."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_((torch.randn(p.shape, generator=g) * 0.02 + (1.0 if "norm" in k and k.endswith("weight") else 0.0)).to(p.device))


def clip_for(tag: str, shape: Iterable[int], seed: int = 0) -> torch.Tensor:
    """a normalised clip batch [B,3,T,H,W] ~ N(0,1) (the reference's own convention for synthetic input, test_efficiency.py:17)"""
    return tensor_for(tag, shape, seed)


def tube_masks(tag: str, batch: int, input_size, mask_ratio: float, generator_cls, seed: int = 0) -> torch.Tensor:
    """bool [B, T'*H'*W'] tube masks from ``generator_cls(input_size, mask_ratio)`` (the reference's TubeMaskingGenerator or this
    repo's), numpy's global RNG seeded from the tag -- one generator call per clip, as the reference's data loader does."""
    np.random.seed(_seed_for(tag, seed))
    gen = generator_cls(input_size, mask_ratio)
    return torch.from_numpy(np.stack([gen() for _ in range(batch)])).bool()
