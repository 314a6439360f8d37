"""Drop-in counterpart of the reference's ``modeling_finetune.py`` operator surface
(class names, constructor arguments, attribute tree, state-dict keys, factory names), backed by
hand-written gfx950 HIP kernels through the C ABI (include/tad_mi355x.h).

What is kept identical to the reference (modeling_finetune.py, file:line in each docstring):
  * module / parameter names and shapes, construction order and init sequence (so the same
    ``torch.manual_seed`` produces the same weights, and checkpoints load unchanged);
  * forward semantics of DropPath / Mlp / Attention / Block / PatchEmbed / VisionTransformer.
What differs: every forward/backward pass runs on the GPU kernels; there is no CPU path -- calling
a module with CPU tensors raises.  ``use_flash_attn`` is accepted and ignored: both of the
reference's attention paths (naive and flash-attn) map onto the one fused attention kernel.
"""
from __future__ import annotations

from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.utils.checkpoint as checkpoint

from . import ops
from ._lib import TadError
from .registry import register_model


def _cfg(url='', **kwargs):
    """modeling_finetune.py:13-20"""
    return {'url': url, 'num_classes': 400, 'input_size': (3, 224, 224), 'pool_size': None, 'crop_pct': .9,
            'interpolation': 'bicubic', 'mean': (0.5, 0.5, 0.5), 'std': (0.5, 0.5, 0.5), **kwargs}


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """timm's trunc_normal_ (uniform -> erfinv -> scale -> clamp); torch.nn.init implements the identical sequence,
    so the RNG stream matches the reference's init (run_inference_simple.py:46-105 holds the same algorithm)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth per sample (modeling_finetune.py:23-34; arithmetic = timm 0.4.12 ``drop_path``:
    mask = floor(keep_prob + U[0,1)), out = x / keep_prob * mask).  ``sample`` returns the per-sample scale
    mask/keep_prob that the fused residual epilogue consumes; ``forced_mask`` lets tests inject the mask."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob
        self.forced_mask = None
        self.presampled = None  # scales drawn for the whole encoder in one launch (VisionTransformer._presample_drop_path)

    def sample(self, batch, device):
        if not self.training or not self.drop_prob:
            return None
        keep = 1.0 - self.drop_prob
        if self.forced_mask is not None:
            mask = self.forced_mask.to(device=device, dtype=torch.float32)
        elif self.presampled:
            return self.presampled.pop(0)
        else:
            mask = (keep + torch.rand(batch, device=device, dtype=torch.float32)).floor_()
        return mask / keep

    def forward(self, x):
        s = self.sample(x.shape[0], x.device)
        if s is None:
            return x
        return x * s.reshape(-1, *([1] * (x.dim() - 1))).to(x.dtype)

    def extra_repr(self) -> str:
        return 'p={}'.format(self.drop_prob)


class Mlp(nn.Module):
    """modeling_finetune.py:37-54"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        if not isinstance(self.act, nn.GELU) or getattr(self.act, "approximate", "none") != "none":
            raise TadError("Mlp: only the exact-erf nn.GELU activation has a fused kernel")

    def forward(self, x):
        if ops.get_precision() == "precise":
            ops._need_gpu(x, "Mlp")
            ops._no_grad_only(x, self.fc1.weight)
            shp = x.shape
            y = ops.precise_mlp(x.detach().float().reshape(-1, shp[-1]).contiguous(), self.fc1.weight, self.fc1.bias, self.fc2.weight,
                                self.fc2.bias)
            return self.drop(y.reshape(*shp[:-1], -1))
        x = ops.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        return self.drop(x)


class Attention(nn.Module):
    """modeling_finetune.py:57-134.  qkv Linear without bias + separate q_bias / v_bias (K bias structurally zero)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., attn_head_dim=None,
                 use_flash_attn=False, causal=False):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if attn_head_dim is not None:
            head_dim = attn_head_dim
        all_head_dim = head_dim * self.num_heads
        self.head_dim = head_dim
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(all_head_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.use_flash_attn = use_flash_attn
        self.causal = causal
        if causal:
            raise TadError("Attention: causal=True is never used by the reference (Block passes causal=False) and has no kernel")

    def _check(self):
        if self.head_dim not in (64, 80):
            raise TadError(f"Attention: head_dim {self.head_dim} has no kernel (64 and 80 do: fused 16-bit MFMA kernels, exact-f32 MFMA kernels in precise mode)")

    def _dropout(self):
        """(p, seed) of attention dropout for this forward (modeling_finetune.py:99-101; the flash path passes dropout_p in training,
        flash_attention_class.py:59-61): active in training only; the seed of the kernels' counter-based mask is drawn from torch's
        default generator, so runs are reproducible under torch.manual_seed.  The reference's jobs all use attn_drop_rate = 0."""
        if self.training and self.attn_drop.p > 0:
            return float(self.attn_drop.p), int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return 0.0, 0

    def forward(self, x):
        self._check()
        if ops.get_precision() == "precise":
            ops._need_gpu(x, "Attention")
            ops._no_grad_only(x, self.qkv.weight)
            if self.training and self.attn_drop.p > 0:
                raise TadError('precision "precise" has no attention dropout (verification runs use attn_drop_rate = 0)')
            B, N, C = x.shape
            y = ops.precise_attention(x.detach().float().reshape(B * N, C).contiguous(), B, N, self.qkv.weight, self.q_bias, self.v_bias,
                                      self.proj.weight, self.proj.bias, self.num_heads, self.scale)
            return self.proj_drop(y.reshape(B, N, -1))
        x = ops.AttentionFn.apply(x, self.qkv.weight, self.q_bias, self.v_bias, self.proj.weight, self.proj.bias, self.num_heads,
                                  self.scale, *self._dropout())
        return self.proj_drop(x)


class Block(nn.Module):
    """modeling_finetune.py:137-166 (pre-LN residual block, optional layer-scale gamma_1/gamma_2)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attn_head_dim=None, use_flash_attn=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                              attn_head_dim=attn_head_dim, use_flash_attn=use_flash_attn, causal=False)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        if init_values > 0:  # (the reference raises on init_values=None here as well)
            self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        else:
            self.gamma_1, self.gamma_2 = None, None

    def _fusable(self):
        return (self.gamma_1 is None and isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm)
                and self.norm1.elementwise_affine and self.norm2.elementwise_affine and self.norm1.eps == self.norm2.eps
                and not (self.training and (self.mlp.drop.p > 0 or self.attn.proj_drop.p > 0 or self.attn.attn_drop.p > 0)))

    def forward(self, x):
        if self._fusable() and ops.get_precision() == "precise":
            self.attn._check()
            if isinstance(self.drop_path, DropPath) and self.training and self.drop_path.drop_prob:
                raise TadError('precision "precise" does not implement drop-path (verification runs use drop_path_rate=0)')
            a, m = self.attn, self.mlp
            return ops.precise_block(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias, a.proj.weight, a.proj.bias,
                                     self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                     a.num_heads, a.scale, self.norm1.eps)
        if self._fusable():
            self.attn._check()
            dp = self.drop_path if isinstance(self.drop_path, DropPath) else None
            s1 = dp.sample(x.shape[0], x.device) if dp is not None else None
            s2 = dp.sample(x.shape[0], x.device) if dp is not None else None
            a, m = self.attn, self.mlp
            return ops.block_apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias, a.proj.weight,
                                   a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight,
                                   m.fc2.bias, s1, s2, a.num_heads, a.scale, self.norm1.eps)
        # layer-scale / dropout variants: composed from the per-operator kernels
        n1 = ops.LayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        n1 = self.attn(n1)
        if self.gamma_1 is not None:
            n1 = self.gamma_1 * n1
        x = x + self.drop_path(n1)
        n2 = ops.LayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        n2 = self.mlp(n2)
        if self.gamma_2 is not None:
            n2 = self.gamma_2 * n2
        return x + self.drop_path(n2)


class PatchEmbed(nn.Module):
    """Tubelet patch embedding (modeling_finetune.py:169-191).  ``proj`` stays an ``nn.Conv3d`` so that the parameter
    names/shapes ([D,C,tub,p,p]) and the default init are the reference's; its arithmetic runs as an im2col + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.tubelet_size = int(tubelet_size)
        num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * (num_frames // self.tubelet_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = num_patches
        self.proj = nn.Conv3d(in_channels=in_chans, out_channels=embed_dim,
                              kernel_size=(self.tubelet_size, patch_size[0], patch_size[1]),
                              stride=(self.tubelet_size, patch_size[0], patch_size[1]))

    # ---- input stage: uint8 frames straight into the patch matrix
    input_mean = (0.485, 0.456, 0.406)  # ImageNet statistics, the values every reference entry point passes
    input_std = (0.229, 0.224, 0.225)   # (run_inference.py:60-61, 76; dota.py:312-314)
    input_bgr = False                   # True: frames come from cv2 (BGR), as in run_inference.py:21
    t_offset = 0                        # ring-buffer start slot (inference.SlidingWindow)

    def set_input_normalization(self, mean, std, bgr=False):
        self.input_mean, self.input_std, self.input_bgr = tuple(float(v) for v in mean), tuple(float(v) for v in std), bool(bgr)

    def _forward_u8(self, frames, pos_embed):
        """frames [B,T,H,W,3] uint8 (decoder layout): normalisation of prepare_image / tensor_normalize fused into the im2col"""
        B, T, H, W, C = frames.shape
        assert C == 3 and H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if ops.get_precision() == "precise":
            raise TadError('precision "precise" takes the normalised f32 clip (the uint8 input stage feeds the bf16 path)')
        return ops.PatchEmbedU8Fn.apply(frames, self.proj.weight, self.proj.bias, pos_embed, self.tubelet_size, self.patch_size[0],
                                        self.input_mean, self.input_std, self.input_bgr, int(self.t_offset))

    def forward(self, x, pos_embed=None, **kwargs):
        if x.dtype == torch.uint8:
            return self._forward_u8(x, pos_embed)
        B, C, T, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if self.patch_size[0] != self.patch_size[1]:
            raise TadError("PatchEmbed: only square patches have a kernel")
        if ops.get_precision() == "precise":
            return ops.precise_patch_embed(x, self.proj.weight, self.proj.bias, pos_embed, self.tubelet_size, self.patch_size[0])
        return ops.PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, pos_embed, self.tubelet_size, self.patch_size[0])


def get_sinusoid_encoding_table(n_position, d_hid):
    """modeling_finetune.py:195-205 -- fp64 numpy table, sin on even / cos on odd columns, cast to fp32, [1,N,D].
    (vectorised; bit-identical to the reference's per-element loops, pinned by tests/golden/g1_bookkeeping.npz)"""
    j = np.arange(d_hid)
    table = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.tensor(table, dtype=torch.float, requires_grad=False).unsqueeze(0)


class VisionTransformer(nn.Module):
    """modeling_finetune.py:208-335.  Extra tolerated kwargs: ``use_mean_pooling`` (VideoMAE-legacy spelling
    used by run_class_finetuning.py:321), ``drop_block_rate`` and unknown ``**kwargs`` are accepted and ignored."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, fc_drop_rate=0., drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, use_flash_attn=True, init_scale=0.,
                 all_frames=16, tubelet_size=2, use_checkpoint=False, final_reduction="fc_norm", use_mean_pooling=None,
                 drop_block_rate=None, **kwargs):
        super().__init__()
        if use_mean_pooling is not None:
            final_reduction = "fc_norm" if use_mean_pooling else "cls"
        self.num_classes = num_classes
        self.num_heads = num_heads
        self.num_features = self.embed_dim = embed_dim
        self.tubelet_size = tubelet_size
        self.num_frames = all_frames
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      num_frames=all_frames, tubelet_size=self.tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.use_checkpoint = use_checkpoint
        if use_learnable_pos_emb:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
            # the forward adds a DETACHED copy (as the reference does, :312-313): the table stays a trainable-looking Parameter that
            # provably never receives a gradient.  The marker keeps it out of the all-reduce bucket counts (parallel.DataParallel)
            # and makes the fused optimizer skip it the way torch.optim skips a ``None`` gradient (optim.FusedAdamW).
            self.pos_embed._tad_never_grad = True
        else:
            # plain tensor attribute, absent from the state dict -- as in the reference (:249-253)
            self.pos_embed = get_sinusoid_encoding_table(num_patches, embed_dim)
        self._pos_dev = None  # device-resident copy (the reference re-uploads the table every forward)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values,
                  use_flash_attn=use_flash_attn) for i in range(depth)])
        assert final_reduction in ("fc_norm", "cls", 'none', None)
        self.final_reduction = final_reduction
        self.norm = nn.Identity() if final_reduction == "fc_norm" else norm_layer(embed_dim)
        self.fc_norm = norm_layer(embed_dim) if final_reduction == "fc_norm" else None
        self.fc_dropout = nn.Dropout(p=fc_drop_rate) if fc_drop_rate > 0 else nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        if use_learnable_pos_emb:
            trunc_normal_(self.pos_embed, std=.02)
        if hasattr(self.head, "weight"):
            trunc_normal_(self.head.weight, std=.02)
        self.apply(self._init_weights)
        if hasattr(self.head, "weight"):
            self.head.weight.data.mul_(init_scale)
            self.head.bias.data.mul_(init_scale)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def _pos_on(self, device):
        if self.pos_embed is None:
            return None
        if isinstance(self.pos_embed, nn.Parameter):
            return self.pos_embed[0]
        if self._pos_dev is None or self._pos_dev.device != device or self._pos_dev.shape != self.pos_embed.shape[1:]:
            self._pos_dev = self.pos_embed[0].to(device=device, dtype=torch.float32).contiguous()
        return self._pos_dev

    def _ln(self, norm, x):
        if isinstance(norm, nn.Identity):
            return x
        return ops.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)

    def forward_features(self, x):
        pos = self._pos_on(x.device)
        if isinstance(self.pos_embed, nn.Parameter):
            # learnable table: the reference adds ``pos_embed.expand(...).clone().detach()`` (modeling_finetune.py:312-313), so the
            # table never receives a gradient -- reproduced (pinned by tests/test_module_cpu.py)
            x = self.patch_embed(x, pos_embed=pos.detach().to(dtype=torch.float32).contiguous())
        else:
            x = self.patch_embed(x, pos_embed=pos)  # fused "+ pos_embed" epilogue
        x = self.pos_drop(x)
        if self.use_checkpoint:
            for blk in self.blocks:
                x = checkpoint.checkpoint(blk, x, use_reentrant=False)
        else:
            self._presample_drop_path(x.shape[0], x.device)
            for blk in self.blocks:
                x = blk(x)
        x = self._ln(self.norm, x)
        if self.final_reduction == "fc_norm":
            return self._ln(self.fc_norm, ops.MeanPoolFn.apply(x))
        elif self.final_reduction == "cls":
            return x[:, 0]
        else:
            return x

    def _presample_drop_path(self, batch, device):
        """All stochastic-depth scales of one forward (two residual branches per block) from ONE torch.rand launch instead of
        ~8 tiny launches per block.  Same distribution as DropPath.sample: floor(keep + U[0,1)) / keep per sample."""
        if not self.training:
            return
        dps = [b.drop_path for b in self.blocks if isinstance(getattr(b, "drop_path", None), DropPath) and b.drop_path.drop_prob
               and b.drop_path.forced_mask is None]
        if not dps:
            return
        # the keep probabilities are constants of the model: built once per device (a host list -> device tensor every forward is a
        # pageable H2D copy that stalls the host until the stream has drained: ~0.2 ms of idle GPU at the start of every step)
        key = (str(device), tuple(d.drop_prob for d in dps))
        if getattr(self, "_keep_cache", (None, None))[0] != key:
            self._keep_cache = (key, torch.tensor([1.0 - d.drop_prob for d in dps], dtype=torch.float32,
                                                  device=device).repeat_interleave(2).unsqueeze(1))
        keep = self._keep_cache[1]
        scales = (keep + torch.rand(2 * len(dps), batch, device=device, dtype=torch.float32)).floor_() / keep
        for i, d in enumerate(dps):
            d.presampled = [scales[2 * i], scales[2 * i + 1]]

    def forward(self, x):
        x = self.forward_features(x)
        x = self.head(self.fc_dropout(x))
        return x


def _vit(embed_dim, depth, num_heads, img_size=224, **kwargs):
    model = VisionTransformer(img_size=img_size, patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                              qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


@register_model
def vit_small_patch16_224(pretrained=False, **kwargs):
    return _vit(384, 12, 6, **kwargs)


@register_model
def vit_base_patch16_224(pretrained=False, **kwargs):
    return _vit(768, 12, 12, **kwargs)


@register_model
def vit_base_patch16_384(pretrained=False, **kwargs):
    return _vit(768, 12, 12, img_size=384, **kwargs)


@register_model
def vit_large_patch16_224(pretrained=False, **kwargs):
    return _vit(1024, 24, 16, **kwargs)


@register_model
def vit_large_patch16_384(pretrained=False, **kwargs):
    return _vit(1024, 24, 16, img_size=384, **kwargs)


@register_model
def vit_large_patch16_512(pretrained=False, **kwargs):
    return _vit(1024, 24, 16, img_size=512, **kwargs)


@register_model
def vit_huge_patch16_224(pretrained=False, **kwargs):
    return _vit(1280, 32, 16, **kwargs)
