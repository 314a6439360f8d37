"""Checkpoint key compatibility between the pre-training and fine-tuning models (run_frame_finetuning.py:399-460;
run_class_finetuning.py has the same block): pick the state dict out of a checkpoint by ``model_key`` ("model|module"), drop a
classifier head of the wrong shape, strip ``backbone.`` / ``encoder.`` prefixes, map ``encoder.norm`` -> ``fc_norm``, interpolate a
learnable positional table to a new grid, then load non-strictly (utils.load_state_dict, utils.py:336-383)."""
from __future__ import annotations

from collections import OrderedDict

import torch


# Key rewrites between a pre-training checkpoint and the fine-tuning model, first match wins: (prefix in the checkpoint, prefix in the
# model).  ``encoder.norm`` must come before ``encoder.``: the MAE encoder's final LayerNorm becomes the classifier's ``fc_norm``.
KEY_PREFIX_MAP = (("backbone.", ""), ("encoder.norm", "fc_norm"), ("encoder.", ""))
HEAD_KEYS = ("head.weight", "head.bias")


def _rename(key: str) -> str:
    for src, dst in KEY_PREFIX_MAP:
        if key.startswith(src):
            return dst + key[len(src):]
    return key


def _select(checkpoint, model_key: str):
    """the state dict inside a checkpoint: the first of ``model_key``'s '|'-separated names present, else the checkpoint itself"""
    if isinstance(checkpoint, dict):
        for name in model_key.split('|'):
            if name in checkpoint:
                return checkpoint[name]
    return checkpoint


def remap_pretrained_state_dict(checkpoint, model: torch.nn.Module, model_key: str = "model|module", num_frames: int = 16):
    own = model.state_dict()
    source = _select(checkpoint, model_key)
    # a classifier head of another shape (other number of classes) is dropped before the renaming, as the reference does
    stale_head = {k for k in HEAD_KEYS if k in source and k in own and source[k].shape != own[k].shape}
    checkpoint_model = OrderedDict((_rename(k), v) for k, v in source.items() if k not in stale_head)
    if 'pos_embed' in checkpoint_model:  # learnable table only (the sinusoid table is not in the state dict)
        pos = checkpoint_model['pos_embed']
        emb = pos.shape[-1]
        num_patches = model.patch_embed.num_patches
        extra = model.pos_embed.shape[-2] - num_patches
        tt = num_frames // model.patch_embed.tubelet_size
        orig = int(((pos.shape[-2] - extra) // tt) ** 0.5)
        new = int((num_patches // tt) ** 0.5)
        if orig != new:
            tok = pos[:, extra:].reshape(-1, tt, orig, orig, emb).reshape(-1, orig, orig, emb).permute(0, 3, 1, 2)
            tok = torch.nn.functional.interpolate(tok, size=(new, new), mode='bicubic', align_corners=False)
            tok = tok.permute(0, 2, 3, 1).reshape(-1, tt, new, new, emb).flatten(1, 3)
            checkpoint_model['pos_embed'] = torch.cat((pos[:, :extra], tok), dim=1)
    return checkpoint_model


def load_state_dict(model: torch.nn.Module, state_dict, prefix: str = '', ignore_missing: str = "relative_position_index"):
    """utils.load_state_dict (utils.py:336-383): non-strict load; returns (missing_keys after the ignore filter, unexpected_keys)"""
    own = model.state_dict()
    filtered = {k[len(prefix):] if prefix and k.startswith(prefix) else k: v for k, v in state_dict.items()}
    res = model.load_state_dict({k: v for k, v in filtered.items() if k in own}, strict=False)
    unexpected = [k for k in filtered if k not in own]
    missing = [k for k in res.missing_keys if not any(ig in k for ig in ignore_missing.split('|'))]
    if missing:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if unexpected:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, unexpected))
    return missing, unexpected
