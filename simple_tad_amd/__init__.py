"""simple_tad_amd -- MI355X-native (gfx950) Video-ViT forward/backward path behind simple-tad's
``modeling_finetune`` operator surface.  Importing the package never touches the GPU; the HIP
shared library is loaded on first use and every op fails loudly if it is missing."""
from . import registry
from .registry import create_model, register_model, list_models
from . import modeling_finetune
from .ops import set_precision, get_precision
from .tuning import TuningScope
from .modeling_finetune import (VisionTransformer, PatchEmbed, Block, Attention, Mlp, DropPath,
                                get_sinusoid_encoding_table)

__all__ = ["set_precision", "get_precision", "TuningScope", "create_model", "register_model", "list_models", "modeling_finetune", "VisionTransformer", "PatchEmbed", "Block",
           "Attention", "Mlp", "DropPath", "get_sinusoid_encoding_table"]
