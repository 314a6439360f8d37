"""Minimal model registry with timm's calling convention.

The reference resolves its models through ``timm.create_model(name, **kw)`` /
``@register_model`` (run_frame_finetuning.py:374-389, modeling_finetune.py:338-398).
When timm is importable the factories are registered there as well, so the reference's
entry scripts pick this implementation up unchanged; otherwise this shim provides
``create_model`` with the one behaviour the scripts rely on: keyword arguments whose
value is ``None`` (``drop_block_rate=None``) are dropped before the factory is called.
"""
from __future__ import annotations

from typing import Callable, Dict

_REGISTRY: Dict[str, Callable] = {}

try:  # pragma: no cover - timm is not installed in the build image
    from timm.models.registry import register_model as _timm_register
except ImportError:
    _timm_register = None


def register_model(fn: Callable) -> Callable:
    _REGISTRY[fn.__name__] = fn
    if _timm_register is not None:  # pragma: no cover
        _timm_register(fn)  # an error here is a real incompatibility with the installed timm: let it surface
    return fn


def list_models():
    return sorted(_REGISTRY)


def create_model(model_name: str, pretrained: bool = False, **kwargs):
    if model_name not in _REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name}); known: {list_models()}")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)
