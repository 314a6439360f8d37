"""Fused AdamW for the fine-tuning step.

Counterpart of ``optim_factory.create_optimizer(..., opt='adamw')`` (optim_factory.py:91-127) over the reference's layer-decay
parameter groups (optim_factory.py:49-88): the same ``param_groups`` surface (``lr``, ``weight_decay``, ``lr_scale``, ``params``)
that ``engine_for_finetuning.train_one_epoch`` rewrites every step (:49-54), the same ``state_dict()`` layout as
``torch.optim.AdamW`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), but ONE HIP launch per step over flat buffers
(``tad_adamw_step``): parameters, gradients and both moments share the layout of ``flat.FlatSpace``; the same pass writes the
bf16 operand copies the next forward reads and the partial sums of g^2 for ``get_grad_norm_`` (utils.py:415-427).
There is no CPU path: the parameters must live on the GPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K
from . import ops
from ._lib import ADAMW_CHUNK, ADAMW_MAX_GROUPS, TadError
from .flat import FlatSpace


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, space: Optional[FlatSpace] = None,
                 mirror: bool = True):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdamW: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        plist = [p for g in self.param_groups for p in g["params"]]
        if not plist or not all(p.is_cuda and p.dtype == torch.float32 for p in plist):
            raise TadError("FusedAdamW runs a HIP kernel over f32 parameters resident on the GPU (no CPU path)")
        if len(self.param_groups) > ADAMW_MAX_GROUPS:
            raise TadError(f"FusedAdamW: {len(self.param_groups)} parameter groups (limit {ADAMW_MAX_GROUPS})")
        self.space = space if space is not None else FlatSpace(plist)
        if not all(p in self.space for p in plist):
            raise TadError("FusedAdamW: the flat layout does not contain every optimised parameter")
        self.flat_grad = self.space.ensure_grads()
        if space is None:
            self.space.install_sinks()  # (DataParallel installs its own, with the bucket notification)
        self.flat_param = self.space.adopt_params()
        self.exp_avg, self.exp_avg_sq = self.space.zeros(), self.space.zeros()
        chunks = self.space.total // ADAMW_CHUNK
        cg = torch.full((chunks,), 255, dtype=torch.uint8)
        # parameters that provably never receive a gradient (``_tad_never_grad``: a learnable pos_embed added detached) are skipped
        # like torch.optim skips a ``None`` gradient: no decay, no moment update, no step count -- their chunks stay unmapped (255)
        self._never = {id(p) for p in plist if getattr(p, "_tad_never_grad", False)}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if id(p) in self._never:
                    continue
                o = self.space.offset[id(p)] // ADAMW_CHUNK
                cg[o:o + FlatSpace.padded(p) // ADAMW_CHUNK] = gi
        self.chunk_group = cg.to(self.space.device)
        self.sumsq_partials = torch.zeros(chunks, dtype=torch.float32, device=self.space.device)
        self.steps = 0
        self._last_updated = None
        self._ragged = False  # True once parameters carry different update counts (some step ran without their gradient)
        for p in plist:  # torch.optim.AdamW's state layout; the moment tensors alias the flat buffers
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.space.view(self.exp_avg, p),
                             "exp_avg_sq": self.space.view(self.exp_avg_sq, p)}
        self.mirror = self.mirror_t = None
        if mirror:
            self.mirror = self.space.zeros(K.operand_dtype())  # operand copies in the 16-bit format of the precision mode in force
            K.cast_op16(self.flat_param, out=self.mirror)
            # transposed copies W^T [K,N] (operand of the input-gradient GEMMs), same offsets, refreshed by ONE batched launch
            self._t_params = [p for p in self.space.params if p.dim() >= 2 and p.shape[0] % 8 == 0 and (p.numel() // p.shape[0]) % 8 == 0]
            if self._t_params:
                self.mirror_t = self.space.zeros(self.mirror.dtype)
                self._t_table = K.transpose_table([(self.space.offset[id(p)], p.shape[0], p.numel() // p.shape[0])
                                                   for p in self._t_params]).to(self.space.device)
                K.transpose_bf16_batched(self.mirror, self.mirror_t, self._t_table)
            self._publish_mirror()

    # ------------------------------------------------------------------ 16-bit operand copies for the forward GEMMs
    def refresh_mirrors(self):
        """re-derive the operand copies from the master weights (after the parameters were changed by anything but this optimizer's
        step: another optimizer over the same flat space, a checkpoint load into ``p.data``) and publish them again"""
        if self.mirror is None:
            return
        K.cast_op16(self.flat_param, out=self.mirror)
        if self.mirror_t is not None:
            K.transpose_bf16_batched(self.mirror, self.mirror_t, self._t_table)
        self._publish_mirror()

    def _publish_mirror(self):
        for p in self.space.params:
            if p.dim() >= 2:
                ops.register_mirror(p, self.space.view(self.mirror, p).view(p.shape[0], -1))
        if self.mirror_t is not None:
            for p in self._t_params:
                o = self.space.offset[id(p)]
                ops.register_mirror(p, self.mirror_t[o:o + p.numel()].view(p.numel() // p.shape[0], p.shape[0]), transposed=True)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[torch.Tensor] = None, want_sumsq: bool = False):
        """One update of every parameter.  ``grad_scale``: optional device scalar multiplied into the gradients (clipping).
        ``want_sumsq``: also return sum(g^2) of the unscaled gradients (device scalar) computed in the same pass."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.space.params_are_flat():
            raise TadError("FusedAdamW: parameters were moved out of the flat buffer (model.to() / .data re-assignment after the "
                           "optimizer was built); rebuild the optimizer")
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for g in self.param_groups:
            if tuple(g["betas"]) != (b1, b2) or g["eps"] != eps:
                raise TadError("FusedAdamW: betas / eps must be the same in every parameter group")
        # torch.optim keeps the update count PER PARAMETER and skips parameters without a gradient (no decay, no moment update,
        # no count).  Fast path: every parameter has a gradient and the same count -> slots == parameter groups and the cached
        # chunk map is used.  Otherwise a slot is a (group, count) pair and the chunk map is rebuilt for this step.
        missing = False
        for p in self.space.params:
            if id(p) in self._never:
                continue
            if p.grad is None:
                missing = True
            else:
                self.space.rehome_grad(p)
        lrs, wds = [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups]
        if not missing and not self._ragged:
            self.steps += 1
            chunk_group, slot_lr, slot_wd, slot_step = self.chunk_group, lrs, wds, [self.steps] * len(lrs)
            updated = [p for p in self.space.params if id(p) not in self._never] if self._never else self.space.params
        else:
            self._ragged = True
            chunk_group = torch.full((self.chunk_group.numel(),), 255, dtype=torch.uint8)
            slots, slot_lr, slot_wd, slot_step, updated = {}, [], [], [], []
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    if p.grad is None or id(p) in self._never:
                        continue
                    t = int(self.state[p]["step"]) + 1
                    si = slots.get((gi, t))
                    if si is None:
                        si = slots[(gi, t)] = len(slot_lr)
                        slot_lr.append(lrs[gi]); slot_wd.append(wds[gi]); slot_step.append(t)
                    o = self.space.offset[id(p)] // ADAMW_CHUNK
                    chunk_group[o:o + FlatSpace.padded(p) // ADAMW_CHUNK] = si
                    updated.append(p)
            if len(slot_lr) > ADAMW_MAX_GROUPS:
                raise TadError("FusedAdamW: too many distinct (group, step) combinations")
            if not updated:
                return loss
            chunk_group = chunk_group.to(self.space.device)
            self.steps += 1
        K.adamw_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, chunk_group, slot_lr, slot_wd, slot_step, b1, b2, eps,
                     param_bf16=self.mirror, grad_scale=grad_scale, sumsq_partials=self.sumsq_partials if want_sumsq else None)
        for p in updated:
            self.state[p]["step"] += 1
        self._last_updated = updated
        ops.invalidate_weight_cache()
        if self.mirror is not None:
            if self.mirror_t is not None:
                K.transpose_bf16_batched(self.mirror, self.mirror_t, self._t_table)
            self._publish_mirror()
        if want_sumsq:
            return self.sumsq_partials.sum()
        return loss

    def rollback_step(self):
        """The last ``step`` turned out to have been skipped ON THE DEVICE (its ``grad_scale`` was 0: the loss scaler found an inf, and
        the kernel left parameters, moments and operand copies untouched): take its update count back, as torch.optim never counted it."""
        if self._last_updated:
            for p in self._last_updated:
                self.state[p]["step"] -= 1
            self.steps -= 1
            self._last_updated = None

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are views into the flat buffer and stay allocated: zero in place (one memset)."""
        self.flat_grad.zero_()
        for p in self.space.params:
            if p.grad is None or p.grad.data_ptr() != self.space.grad_view(p).data_ptr():
                p.grad = self.space.grad_view(p)

    # ------------------------------------------------------------------ checkpoints (utils.save_model / auto_load_model)
    def load_state_dict(self, state_dict):
        views = {id(p): (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in self.space.params}
        super().load_state_dict(state_dict)  # replaces the state tensors with copies of the loaded ones
        steps = 0
        for p in self.space.params:
            st = self.state.get(p)
            if not st:
                continue
            ea, eas = views[id(p)]
            ea.copy_(st["exp_avg"])
            eas.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = ea, eas
            st["step"] = torch.as_tensor(float(st["step"]))
            steps = max(steps, int(st["step"]))
        self.steps = steps
        self._ragged = len({int(self.state[p]["step"]) for p in self.space.params if self.state.get(p) and id(p) not in self._never}) > 1
