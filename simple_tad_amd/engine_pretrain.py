"""Counterpart of ``engine_for_pretraining.train_one_epoch`` (engine_for_pretraining.py:16-152) for the MAE pre-training path
(SURVEY 8f-2): per step -- lr / weight-decay assignment (:39-45), reconstruction target from the clip (:51-66, one HIP kernel on
the masked tokens only), model forward, ``nn.MSELoss`` (:68-70, fused loss + gradient kernel), ``loss.item()``, scaler step with
``clip_grad=max_norm`` (:79-81), synchronise.  Left out: the per-head gradient-norm diagnostics (:30-33, 82-89, 134-147), the
``gc.collect()/empty_cache()`` per step (:36-37), tensorboard logging."""
from __future__ import annotations

import math
import sys
from typing import Iterable

import torch

from . import kernels as K
from . import ops
from .modeling_pretrain import token_indices
from .parallel import DataParallel

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)  # timm.data.constants, engine_for_pretraining.py:10
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def reconstruction_target(videos: torch.Tensor, bool_masked_pos: torch.Tensor, patch_size: int = 16, tubelet_size: int = 2,
                          normlize_target: bool = True, num_masked=None) -> torch.Tensor:
    """labels [B, N_mask, tub*p*p*3] (engine_for_pretraining.py:51-66)"""
    _, mask_tok = token_indices(bool_masked_pos.flatten(1), num_masked)
    return K.mae_target(videos.float().contiguous(), mask_tok.reshape(-1), tubelet_size, patch_size, IMAGENET_DEFAULT_MEAN,
                        IMAGENET_DEFAULT_STD, normlize_target)


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer, device: torch.device, epoch: int, loss_scaler,
                    max_norm: float = 0, patch_size: int = 16, normlize_target: bool = True, start_steps=0, lr_schedule_values=None,
                    wd_schedule_values=None, tubelet_size: int = 2, log=None):
    model.train()
    dp = model if isinstance(model, DataParallel) else None
    inner = dp.module if dp is not None else model
    zero = dp.zero_grad if dp is not None else (lambda: optimizer.zero_grad(set_to_none=False))
    params = [p for p in model.parameters() if p.requires_grad]
    names = ("loss", "grad_norm", "lr", "min_lr", "loss_scale", "weight_decay")  # fixed list: every rank packs the same rows
    stats = {k: [] for k in names}
    for step, batch in enumerate(data_loader):
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group.get("lr_scale", 1.0)
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        videos, bool_masked_pos = batch[0], batch[1]
        num_masked = int(torch.as_tensor(bool_masked_pos)[0].sum())  # host-side count (the mask comes from the loader's generator)
        videos = videos.to(device, non_blocking=True)
        bool_masked_pos = torch.as_tensor(bool_masked_pos).to(device, non_blocking=True).flatten(1).to(torch.bool)
        with torch.no_grad():
            labels = reconstruction_target(videos, bool_masked_pos, patch_size, tubelet_size, normlize_target, num_masked)
        outputs = inner(videos, bool_masked_pos, num_masked=num_masked) if dp is None else dp(videos, bool_masked_pos, num_masked=num_masked)
        loss = ops.MseLossFn.apply(outputs, labels)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        zero()
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm if max_norm else None, parameters=params)
        if device.type == "cuda":
            torch.cuda.synchronize()
        stats["loss"].append(loss_value)
        stats["grad_norm"].append(None if grad_norm is None else float(grad_norm))
        stats["loss_scale"].append(loss_scaler.state_dict()["scale"])
        stats["lr"].append(max(g["lr"] for g in optimizer.param_groups))
        stats["min_lr"].append(min(g["lr"] for g in optimizer.param_groups))
        stats["weight_decay"].append(next((g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0), None))
        if log is not None:
            log(epoch, step, stats)
    # engine_for_pretraining.py:149-152: metric_logger.synchronize_between_processes() -> {k: meter.global_avg}
    from .engine import synchronize_meters
    stats["averaged"] = synchronize_meters(stats, device, group=dp.pg if dp is not None else None, names=names)
    return stats
