"""One flat layout shared by the gradient exchange (parallel.DataParallel) and the fused optimizer (optim.FusedAdamW).

All trainable parameters are laid out in ONE contiguous buffer per role (gradients, parameters, Adam moments, bf16 operand
mirror) with identical offsets, in reverse registration order (head first, patch-embed last = roughly the order in which
backward produces gradients, so all-reduce buckets complete early).  Every tensor starts on a 4096-element boundary
(``_lib.ADAMW_CHUNK``): a chunk of the fused AdamW kernel then belongs to exactly one parameter group, and every view is
16-KiB aligned for the vector loads of the HIP kernels.  Padding elements stay zero in every buffer.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ._lib import ADAMW_CHUNK


class FlatSpace:
    ALIGN = ADAMW_CHUNK

    def __init__(self, params: List[torch.nn.Parameter], reverse: bool = True):
        params = [p for p in params if p.requires_grad]
        assert params, "no trainable parameters"
        self.device, self.dtype = params[0].device, params[0].dtype
        assert all(p.device == self.device and p.dtype == self.dtype for p in params), "parameters must share device and dtype"
        assert len({id(p) for p in params}) == len(params), "duplicate parameters"
        self.params = list(reversed(params)) if reverse else list(params)
        self.offset: Dict[int, int] = {}
        total = 0
        for p in self.params:
            self.offset[id(p)] = total
            total += self.padded(p)
        self.total = total
        self.flat_grad: Optional[torch.Tensor] = None
        self.flat_param: Optional[torch.Tensor] = None
        self._grad_views: Dict[int, torch.Tensor] = {}

    @classmethod
    def padded(cls, p) -> int:
        return (p.numel() + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def __contains__(self, p) -> bool:
        return id(p) in self.offset

    def view(self, flat: torch.Tensor, p) -> torch.Tensor:
        o = self.offset[id(p)]
        return flat[o:o + p.numel()].view(p.shape)

    def zeros(self, dtype=None) -> torch.Tensor:
        return torch.zeros(self.total, dtype=dtype or self.dtype, device=self.device)

    # -- gradients: p.grad become views into flat_grad (existing gradients are kept)
    def ensure_grads(self) -> torch.Tensor:
        if self.flat_grad is None:
            self.flat_grad = self.zeros()
            for p in self.params:
                v = self.view(self.flat_grad, p)
                if p.grad is not None:
                    v.copy_(p.grad)
                self._grad_views[id(p)] = v
                p.grad = v
        return self.flat_grad

    def install_sinks(self, notify=None) -> None:
        """Let the weight-gradient kernels accumulate straight into the flat gradient buffer (ops.register_grad_sink); ``notify(p)``
        is called after a gradient has been written this way (autograd's own hooks do not fire for it)."""
        from . import ops
        self.ensure_grads()
        for p in self.params:
            if p.is_cuda:
                ops.register_grad_sink(p, self._grad_views[id(p)], notify)

    def grad_view(self, p) -> torch.Tensor:
        return self._grad_views[id(p)]

    def rehome_grad(self, p) -> None:
        """After ``zero_grad(set_to_none=True)`` autograd allocates a fresh gradient tensor: copy it back into the flat buffer."""
        v = self._grad_views[id(p)]
        if p.grad is not v and (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
            p.grad = v

    # -- parameters: p.data become views into flat_param (values preserved)
    def adopt_params(self) -> torch.Tensor:
        if self.flat_param is None:
            self.flat_param = self.zeros()
            with torch.no_grad():
                for p in self.params:
                    v = self.view(self.flat_param, p)
                    v.copy_(p.detach())
                    p.data = v
        return self.flat_param

    def params_are_flat(self) -> bool:
        if self.flat_param is None:
            return False
        base = self.flat_param.data_ptr()
        es = self.flat_param.element_size()
        return all(p.data_ptr() == base + self.offset[id(p)] * es for p in self.params)
