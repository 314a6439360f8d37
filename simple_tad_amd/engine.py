"""Thin counterparts of the training-runtime pieces the fine-tune hot loop needs
(reference: engine_for_finetuning.py:24-140, utils.py:386-447, optim_factory.py:24-127).

Same step ordering as ``engine_for_finetuning.train_one_epoch``: per-step lr/wd assignment from the
schedule tables, forward, loss, ``loss.item()`` finiteness check, backward, gradient norm / clipping,
optimizer step, ``zero_grad``, device synchronise, meters.  Differences, by design: loss scaling is live only
for IEEE-half operands (``set_precision("half")``, the reference's own autocast dtype); the bfloat16 kernels need
none and the scaler then reports ``scale == 1.0``; the gradient all-reduce is the bucketed RCCL exchange of
``parallel.DataParallel``.
"""
from __future__ import annotations

import json
import math
import sys
from typing import Iterable, Optional

import numpy as np
import torch

from . import kernels as K
from . import ops
from .parallel import DataParallel


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """utils.cosine_scheduler (utils.py:430-447): linear warm-up then half-cosine, one value per iteration."""
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    cos = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n)) for i in range(n)])
    schedule = np.concatenate((warm, cos))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def get_num_layer_for_vit(var_name, num_max_layer):
    """optim_factory.py:24-35"""
    if var_name in ("cls_token", "mask_token", "pos_embed") or var_name.startswith("patch_embed"):
        return 0
    if var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    if var_name.startswith("blocks"):
        return int(var_name.split('.')[1]) + 1
    return num_max_layer - 1


class LayerDecayValueAssigner:
    """optim_factory.py:38-46; values = [decay ** (L + 1 - i) for i in range(L + 2)] (run_class_finetuning.py:430-434)"""

    def __init__(self, values):
        self.values = values

    @classmethod
    def from_decay(cls, layer_decay: float, num_layers: int):
        return cls([layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)])

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None, verbose=False):
    """The grouping rule of optim_factory.get_parameter_groups (optim_factory.py:49-88): a parameter is undecayed when it is 1-D,
    a ``.bias`` or skip-listed; with layer-wise lr decay every (layer, decayed?) pair is a group of its own carrying ``lr_scale``.
    Groups come out in order of first appearance in ``model.named_parameters()`` -- the order the reference hands to AdamW, and what
    golden G12 pins (group names, member order, weight decay, lr_scale)."""
    skip = frozenset(skip_list)
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    # key of a parameter = (layer id | None, decayed?)
    keys = [(None if get_num_layer is None else get_num_layer(n), not (p.dim() == 1 or n.endswith(".bias") or n in skip)) for n, p in trainable]
    groups, members = {}, {}
    for key, (name, param) in zip(keys, trainable):
        if key not in groups:
            layer, decayed = key
            groups[key] = {"weight_decay": weight_decay if decayed else 0., "params": [],
                           "lr_scale": 1. if get_layer_scale is None else get_layer_scale(layer)}
            members[key] = []
        groups[key]["params"].append(param)
        members[key].append(name)
    if verbose:
        def label(key):
            kind = "decay" if key[1] else "no_decay"
            return kind if key[0] is None else "layer_%d_%s" % (key[0], kind)
        print("Param groups = %s" % json.dumps({label(k): dict(groups[k], params=members[k]) for k in groups}, indent=2))
    return list(groups.values())


def create_optimizer(model, lr, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, layer_decay: Optional[float] = None,
                     skip_list=None, fused_kernel: Optional[bool] = None):
    """AdamW over the reference's parameter groups (optim_factory.create_optimizer with opt='adamw',
    optim_factory.py:91-127); the rest of the reference's optimizer zoo is out of scope.  On the GPU this is
    optim.FusedAdamW (one HIP launch per step over the flat layout it shares with DataParallel); ``fused_kernel=False`` or
    CPU parameters give torch.optim.AdamW over the same groups."""
    inner = model.module if isinstance(model, DataParallel) else model
    skip = set(skip_list) if skip_list is not None else (inner.no_weight_decay() if hasattr(inner, "no_weight_decay") else set())
    assigner = LayerDecayValueAssigner.from_decay(layer_decay, inner.get_num_layers()) if layer_decay and layer_decay < 1.0 else None
    groups = get_parameter_groups(inner, weight_decay, skip, assigner.get_layer_id if assigner else None,
                                  assigner.get_scale if assigner else None)
    on_gpu = all(p.is_cuda for g in groups for p in g["params"])
    if fused_kernel is None:
        fused_kernel = on_gpu
    if fused_kernel:
        from .optim import FusedAdamW
        return FusedAdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=0.0,
                          space=model.space if isinstance(model, DataParallel) else None)
    return torch.optim.AdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=0.0, fused=on_gpu)


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """utils.get_grad_norm_ (utils.py:415-427) for norm_type 2: norm(stack(norm(g))) == sqrt(sum ||g||^2).
    On GPU the sum of squares runs through tad_sumsq_f32 (one pass over the gradients)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    assert float(norm_type) == 2.0, "only the L2 norm is used by the reference's engines"
    if grads[0].is_cuda:
        acc = torch.zeros(1, dtype=torch.float32, device=grads[0].device)
        for g in grads:
            K.sumsq(g.detach().reshape(-1) if g.is_contiguous() else g.detach().contiguous().reshape(-1), acc)
        return acc.sqrt()[0]
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)


class NativeScalerWithGradNormCount:
    """utils.NativeScalerWithGradNormCount (utils.py:386-412): backward -> (all-reduce wait) -> unscale -> clip or norm -> optimizer
    step -> scale update.  The reference wraps ``torch.cuda.amp.GradScaler()`` because its autocast arithmetic is float16; here loss
    scaling is live exactly when the kernels run on IEEE-half operands (``set_precision("half")``) and a no-op (scale 1.0) for the
    bfloat16 and precise modes, whose range is f32's.  GradScaler's defaults and rules: scale 65536, halved after a step whose
    gradients held an inf / NaN (that step is skipped), doubled after 2000 consecutive good steps.  The scale is removed inside the
    fused AdamW (its ``grad_scale`` operand, together with the clipping coefficient), so no extra pass touches the gradients.
    GradScaler.step reads "found inf" on the host before it calls the optimizer -- a host sync between backward and the optimizer that
    leaves the GPU idle while the host queues the next launches.  With the fused AdamW the decision is taken ON THE DEVICE instead
    (``*grad_scale == 0`` skips the update inside the kernel) and the host reads the flag one step late, at the start of the next
    call -- before the scale is used again, so scale, skipped steps and the optimizer's step counts end up exactly as GradScaler's."""
    state_dict_key = "amp_scaler"

    def __init__(self, model: Optional[torch.nn.Module] = None, enabled: Optional[bool] = None, init_scale: float = 65536.0,
                 growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        self.model = model
        self.enabled = enabled  # None: follow the precision mode at call time
        self.scale = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.growth_tracker = 0
        self.skipped_steps = 0
        self._pending = None  # (found-inf flag in pinned host memory, the event behind its copy, optimizer) of the last fused step, not yet read
        self._flag_host = None

    def _settle(self):
        """read the previous fused step's found-inf flag and bring the scale and, after a skipped step, the optimizer's step counts up
        to date.  The flag travelled to pinned host memory by an asynchronous copy issued right behind the norm kernel, and an event
        recorded there is what the host waits for -- NOT the stream: by the time this runs the forward of the next step is already
        queued, so the GPU keeps working through the wait.  (Round 4 read the flag with ``.item()``: a copy queued BEHIND that forward,
        i.e. a full stream drain in front of every backward pass -- 0.5-1.0 ms of idle GPU per step in the kernel trace,
        profiles/r05_kernel_stats_half.txt.)"""
        if self._pending is None:
            return
        flag, event, optimizer, scaled = self._pending
        self._pending = None
        if event is not None:
            event.synchronize()
        found_inf = bool(flag[0].item() != 0)  # (pinned host memory behind its event, or a CPU tensor: no device round trip)
        if scaled:
            self._update_scale(found_inf)
        elif found_inf:
            self.skipped_steps += 1  # (clipping without loss scaling: the step was skipped on the device, the scale is not in play)
        if found_inf:
            optimizer.rollback_step()

    def scaling(self) -> bool:
        return (ops.get_precision() == "half") if self.enabled is None else bool(self.enabled)

    def _update_scale(self, found_inf: bool):
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
            self.skipped_steps += 1
        else:
            self.growth_tracker += 1
            if self.growth_tracker >= self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        dp = self.model if isinstance(self.model, DataParallel) else None
        if dp is not None:
            dp.require_sync = bool(update_grad)  # gradient accumulation: exchange only on the last micro-step
        scaling = self.scaling()
        self._settle()
        (loss * self.scale if scaling else loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        if dp is not None:
            dp.finish()
        from .optim import FusedAdamW
        inv = 1.0 / self.scale if scaling else 1.0
        if isinstance(optimizer, FusedAdamW):
            # the norm of utils.get_grad_norm_ comes out of the optimizer's own pass over the gradients; with clipping or loss scaling
            # the coefficient (1 / scale) * min(1, max_norm / (norm + 1e-6)) of GradScaler.unscale_ + clip_grad_norm_ stays on the device
            clip = clip_grad is not None and clip_grad > 0
            if clip or scaling:
                # ONE pass over the flat gradients + a one-thread finish: norm, coefficient (0 = overflow: the kernel skips the step)
                # and the found-inf flag, all on the device (tad_grad_norm_coef)
                # A NON-FINITE norm makes the coefficient 0 in EVERY mode (tad_grad_norm_coef), i.e. the update is skipped on the device.
                # Under loss scaling that is GradScaler's rule.  Under clipping WITHOUT scaling (bf16 / precise) it is a deliberate
                # difference from the reference, whose clip_grad_norm_ would multiply the gradients by NaN and write NaN into every weight
                # (utils.py:401-404): here the weights survive, and the skip is accounted for like a scaled one -- counted in
                # `skipped_steps`, the optimizer's step counts rolled back -- when the flag is read at the next call.
                nc = K.grad_norm_coef(optimizer.flat_grad, inv, clip_grad if clip else 0.0)
                norm = nc[0]
                # the flag is read at the next call, from pinned host memory behind an event (see _settle)
                if self._flag_host is None:
                    self._flag_host = torch.zeros(1, dtype=torch.float32).pin_memory()
                self._flag_host.copy_(nc[2:3], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (self._flag_host, ev, optimizer, scaling)
                optimizer.step(grad_scale=nc[1:2])
            else:
                norm = optimizer.step(want_sumsq=True).sqrt()
            return norm
        if scaling:  # any other optimizer: unscale in place (GradScaler.unscale_), then the reference's order
            if parameters is None:
                raise ValueError("NativeScalerWithGradNormCount: loss scaling with an optimizer other than FusedAdamW needs `parameters` "
                                 "(the gradients are unscaled and checked here before the step)")
            grads = [p.grad for p in parameters if p.grad is not None]
            found = torch.zeros((), dtype=torch.float32, device=grads[0].device) if grads else None
            for g in grads:  # one reduction per tensor on the device, ONE host read for all of them
                g.mul_(inv)
                found += (~torch.isfinite(g)).any().float()
            found_inf = bool(grads) and bool(found.item() > 0)
            self._update_scale(found_inf)
            if found_inf:
                return torch.tensor(float("inf"))
        if clip_grad is not None and clip_grad > 0:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            if dp is not None and not scaling:
                acc = torch.zeros(1, dtype=torch.float32, device=dp.flat_grad.device)
                norm = (K.sumsq(dp.flat_grad, acc).sqrt()[0] if dp.flat_grad.is_cuda else dp.flat_grad.norm())
            else:
                norm = get_grad_norm_(parameters)
        optimizer.step()
        ops.invalidate_weight_cache()  # fused optimizers do not bump Parameter._version (see ops._cached)
        return norm

    def state_dict(self):
        """GradScaler.state_dict()'s keys.  ``scale`` reads 1.0 while no scaling is applied -- the engines log it as loss_scale each step
        (engine_for_finetuning.py:100) -- and the scale this object carries travels beside it as ``_scale`` so that a checkpoint
        written in one precision mode resumes correctly in another (the 65536 of a half-mode run used to be dropped when
        the precision was set after the load, and a bf16-mode checkpoint installed scale 1.0 in a half-mode run)."""
        self._settle()
        return {"scale": self.scale if self.scaling() else 1.0, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self.growth_tracker, "_scale": self.scale}

    def load_state_dict(self, state_dict):
        """restores the carried scale whatever the precision mode is at this moment; a checkpoint without ``_scale`` (the reference's
        GradScaler, or an older run of this package) provides ``scale``, which is ignored when it is the 1.0 placeholder of a run that
        applied no scaling"""
        if "_scale" in state_dict:
            self.scale = float(state_dict["_scale"])
        elif "scale" in state_dict and float(state_dict["scale"]) != 1.0:
            self.scale = float(state_dict["scale"])
        elif "scale" in state_dict and self.scaling():
            import warnings
            warnings.warn("NativeScalerWithGradNormCount.load_state_dict: the checkpoint carries loss scale 1.0 (written without loss scaling); "
                          f"keeping this run's scale {self.scale:g} for the IEEE-half mode")
        self.growth_tracker = int(state_dict.get("_growth_tracker", self.growth_tracker))


METER_NAMES = ("loss", "class_acc", "grad_norm", "lr", "min_lr", "loss_scale")


def synchronize_meters(stats: dict, device=None, group=None, names=None) -> dict:
    """C4 of SURVEY 2.3 -- ``MetricLogger.synchronize_between_processes`` (utils.py:71-82, called at engine_for_finetuning.py:138):
    every meter's (count, total) summed over the ranks, returned as ``{name: global_avg}`` (engine_for_finetuning.py:140), so every
    rank logs the same epoch averages.  The reference issues one barrier + one fp64 all-reduce PER meter; here all meters travel in
    ONE fp64 all-reduce of a [n_meters, 2] tensor over ``group`` (the process group of the gradient exchange; default group if
    None).  ``names`` fixes the meter list: every rank must pack the same rows in the same order, so the engines pass their full
    list and a meter a rank never updated travels as (0, 0) -- a per-rank list derived from the dict would make the collective's
    shape differ across ranks and hang.  Single process: the local averages.  ``None`` entries are skipped as MetricLogger.update
    does (utils.py:121-122)."""
    import torch.distributed as dist
    if names is None:
        names = sorted(k for k, v in stats.items() if isinstance(v, list))
    names = list(names)
    packed = torch.zeros((len(names), 2), dtype=torch.float64)
    for i, k in enumerate(names):
        vals = [float(v) for v in stats.get(k, ()) if v is not None]
        packed[i, 0], packed[i, 1] = len(vals), sum(vals)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            packed = packed.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(packed, group=group)
        packed = packed.cpu()
    return {k: (packed[i, 1] / packed[i, 0]).item() for i, k in enumerate(names) if packed[i, 0] > 0}


def train_class_batch(model, samples, target, criterion):
    """engine_for_finetuning.py:13-16"""
    outputs = model(samples)
    loss = criterion(outputs, target)
    return loss, outputs


def train_one_epoch(model: torch.nn.Module, criterion, data_loader: Iterable, optimizer, device, epoch: int, loss_scaler,
                    max_norm: float = 0, start_steps=0, lr_schedule_values=None, wd_schedule_values=None,
                    num_training_steps_per_epoch=None, update_freq=1, log=None):
    """engine_for_finetuning.train_one_epoch (engine_for_finetuning.py:24-140) without mixup / EMA / DeepSpeed branches.
    The loader yields (samples [B,3,T,H,W], targets [B], *rest).  Returns the per-step meter lists of THIS rank plus
    ``stats["averaged"]`` = the cross-rank epoch averages the reference returns (``{k: meter.global_avg}``)."""
    model.train(True)
    dp = model if isinstance(model, DataParallel) else None
    zero = (dp.zero_grad if dp is not None else lambda: optimizer.zero_grad(set_to_none=False))
    zero()
    stats = {k: [] for k in METER_NAMES}
    params = [p for p in model.parameters() if p.requires_grad]
    for data_iter_step, batch in enumerate(data_loader):
        samples, targets = batch[0], batch[1]
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if (lr_schedule_values is not None or wd_schedule_values is not None) and data_iter_step % update_freq == 0:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group.get("lr_scale", 1.0)
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        loss, output = train_class_batch(model, samples, targets, criterion)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        loss = loss / update_freq
        last = (data_iter_step + 1) % update_freq == 0
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm if max_norm else None, parameters=params, update_grad=last)
        if last:
            zero()
        if device.type == "cuda":
            torch.cuda.synchronize()
        stats["loss"].append(loss_value)
        stats["class_acc"].append((output.max(-1)[-1] == targets).float().mean().item())
        stats["grad_norm"].append(None if grad_norm is None else float(grad_norm))
        stats["loss_scale"].append(loss_scaler.state_dict()["scale"])
        stats["lr"].append(max(g["lr"] for g in optimizer.param_groups))
        stats["min_lr"].append(min(g["lr"] for g in optimizer.param_groups))
        if log is not None:
            log(epoch, data_iter_step, stats)
    # gather the stats from all processes (engine_for_finetuning.py:137-140): per-step lists stay per rank, "averaged" is global
    stats["averaged"] = synchronize_meters(stats, device, group=dp.pg if dp is not None else None, names=METER_NAMES)
    return stats


# ----------------------------------------------------------------------------------------------------------------- evaluation
def gather_predictions(tensors, world_size=None):
    """utils.gather_predictions_nontensor (utils.py:791-810) for lists of per-batch tensors: concatenate locally, all-gather the
    (equal-shaped or padded) result across ranks; single process: the local concatenation."""
    import torch.distributed as dist
    local = torch.cat([t.detach() for t in tensors], dim=0)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s) for s in sizes))
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: int(s)] for o, s in zip(out, sizes)], dim=0)


@torch.no_grad()
def validation_one_epoch(data_loader, model, device):
    """engine_for_frame_finetuning.validation_one_epoch (:282-383) without the TTC / smoothed-label / plotting branches: eval-mode
    forward over the loader, CE loss and top-1 per batch, predictions gathered across ranks, then ``metrics.calculate_metrics``."""
    from . import metrics as M
    criterion = torch.nn.CrossEntropyLoss()
    model.eval()
    preds, labels, losses, correct, seen = [], [], [], 0, 0
    for batch in data_loader:
        videos = batch[0].to(device, non_blocking=True)
        target = batch[1].to(device, non_blocking=True)
        output = model(videos)
        losses.append(criterion(output, target).item())
        correct += int((output.max(-1)[1] == target).sum())
        seen += videos.shape[0]
        preds.append(output.detach())
        labels.append(target.detach())
    all_preds, all_labels = gather_predictions(preds), gather_predictions(labels)
    acc, recall, precision, f1, confmat, auroc, ap, pr_curve, roc_curve, mcc = M.calculate_metrics(all_preds, all_labels)
    values = torch.nn.functional.softmax(all_preds, dim=1)[:, 1]
    my = {"metr_acc": acc, "recall": recall, "precision": precision, "f1": f1, "auroc": auroc, "ap": ap, "mcc_auc": mcc[0],
          "mcc_max": mcc[1], "mcc_max_thresh": mcc[2], "mcc_05": mcc[3],
          "logitsP_mean": all_preds[:, 1].mean().item(), "logitsP_std": all_preds[:, 1].std().item(),
          "logitsP_median": all_preds[:, 1].median().item(), "logitsN_mean": all_preds[:, 0].mean().item(),
          "logitsN_std": all_preds[:, 0].std().item(), "logitsN_median": all_preds[:, 0].median().item(),
          "probs_mean": values.mean().item(), "probs_std": values.std().item(), "probs_median": values.median().item()}
    return {"loss": float(np.mean(losses)), "acc": 100.0 * correct / max(seen, 1)}, my, {"confmat": confmat, "pr_curve": pr_curve, "roc_curve": roc_curve}
