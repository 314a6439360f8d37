"""Drop-in counterpart of the reference's ``modeling_pretrain.py`` (VideoMAE masked-autoencoder pre-training):
same class names, constructor arguments, attribute tree, state-dict keys, init sequence and factory names; forward and
backward run on the HIP kernels (shared Block stack of ``modeling_finetune`` + token gather / decoder-input assembly kernels).

Reference lines are cited per class (modeling_pretrain.py).  Differences, by design:
  * ``forward(x, mask, num_masked=None)``: ``num_masked`` (masked tokens per clip, identical for every clip as the reference's
    ``reshape(B, -1, C)`` requires) may be passed to avoid one device sync; otherwise it is read from ``mask[0]``.
  * There is no CPU path.  head_dim 64 (small / base / large) and head_dim 80 (huge) run the fused 16-bit MFMA attention kernels.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn
import torch.utils.checkpoint as checkpoint

from . import ops
from ._lib import TadError
from .modeling_finetune import Block, PatchEmbed, _cfg, get_sinusoid_encoding_table
from .registry import register_model

__all__ = ['pretrain_videomae_small_patch16_224', 'pretrain_videomae_base_patch16_224', 'pretrain_videomae_large_patch16_224',
           'pretrain_videomae_huge_patch16_224']


def trunc_normal_(tensor, mean=0., std=1.):
    """modeling_pretrain.py:14-15: truncation at +-std (not +-2)"""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)


def _init_weights(m):
    """modeling_pretrain.py:66-73 (same text at :156-163, :253-260)"""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if isinstance(m, nn.Linear) and m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


def token_indices(mask: torch.Tensor, num_masked=None):
    """bool mask [B,N] (True = masked) -> (vis_tok [B,Nv], mask_tok [B,Nm]) int32 token indices in ascending order per clip --
    the order boolean indexing ``x[~mask]`` / ``x[mask]`` produces (modeling_pretrain.py:98, 285-286).  A stable argsort of the
    mask does it without the device sync of ``nonzero``."""
    if mask.dtype != torch.bool:
        mask = mask.to(torch.bool)
    B, N = mask.shape
    if num_masked is None:
        num_masked = int(mask[0].sum())
    order = torch.argsort(mask.to(torch.uint8), dim=1, stable=True).to(torch.int32)
    nv = N - int(num_masked)
    return order[:, :nv].contiguous(), order[:, nv:].contiguous()


class PretrainVisionTransformerEncoder(nn.Module):
    """modeling_pretrain.py:27-113"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 init_values=None, tubelet_size=2, use_checkpoint=False, use_learnable_pos_emb=False, use_flash_attn=True):
        super().__init__()
        self.num_classes = num_classes
        self.num_heads = num_heads
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.use_checkpoint = use_checkpoint
        if use_learnable_pos_emb:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.pos_embed._tad_never_grad = True  # added detached (:96 of the reference): see modeling_finetune.VisionTransformer
        else:
            self.pos_embed = get_sinusoid_encoding_table(num_patches, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values,
                  use_flash_attn=use_flash_attn)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        if use_learnable_pos_emb:
            trunc_normal_(self.pos_embed, std=.02)
        self.apply(_init_weights)
        self._pos_dev = None

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
        if self._pos_dev is None or self._pos_dev.device != device:
            self._pos_dev = self.pos_embed[0].detach().to(device=device, dtype=torch.float32).contiguous()
        return self._pos_dev

    def forward_features(self, x, mask, num_masked=None, vis_tok=None):
        if isinstance(self.pos_embed, nn.Parameter):
            x = self.patch_embed(x)
            x = x + self.pos_embed.type_as(x).to(x.device).clone().detach()  # [1,N+1,D] vs [B,N,D]: fails as in the reference (:96)
        else:
            x = self.patch_embed(x, pos_embed=self._pos_on(x.device))  # "+ pos_embed" fused into the patch-embed GEMM epilogue
        B, N, C = x.shape
        if vis_tok is None:
            vis_tok, _ = token_indices(mask, num_masked)
        rows = (vis_tok + (torch.arange(B, device=x.device, dtype=torch.int32) * N).unsqueeze(1)).reshape(-1).contiguous()
        x_vis = ops.GatherRowsFn.apply(x, rows, B)  # x[~mask].reshape(B, -1, C)
        if self.use_checkpoint:
            for blk in self.blocks:
                x_vis = checkpoint.checkpoint(blk, x_vis, use_reentrant=False)
        else:
            for blk in self.blocks:
                x_vis = blk(x_vis)
        return ops.LayerNormFn.apply(x_vis, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x, mask, num_masked=None, vis_tok=None):
        x = self.forward_features(x, mask, num_masked, vis_tok)
        if isinstance(self.head, nn.Identity):
            return x
        return ops.LinearFn.apply(x, self.head.weight, self.head.bias)


class PretrainVisionTransformerDecoder(nn.Module):
    """modeling_pretrain.py:115-182"""

    def __init__(self, patch_size=16, num_classes=768, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=None,
                 num_patches=196, tubelet_size=2, use_checkpoint=False, use_flash_attn=True):
        super().__init__()
        self.num_classes = num_classes
        assert num_classes == 3 * tubelet_size * patch_size ** 2
        self.num_features = self.embed_dim = embed_dim
        self.patch_size = patch_size
        self.use_checkpoint = use_checkpoint
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values,
                  use_flash_attn=use_flash_attn)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(_init_weights)

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

    def forward(self, x, return_token_num):
        if self.use_checkpoint:
            for blk in self.blocks:
                x = checkpoint.checkpoint(blk, x, use_reentrant=False)
        else:
            for blk in self.blocks:
                x = blk(x)
        if return_token_num > 0:
            x = x[:, -return_token_num:]  # only the mask tokens predict pixels
        x = ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.head, nn.Identity):
            return x
        return ops.LinearFn.apply(x, self.head.weight, self.head.bias)


class PretrainVisionTransformer(nn.Module):
    """modeling_pretrain.py:184-291"""

    def __init__(self, img_size=224, patch_size=16, encoder_in_chans=3, encoder_num_classes=0, encoder_embed_dim=768, encoder_depth=12,
                 encoder_num_heads=12, decoder_num_classes=1536, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=8,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 init_values=0., use_learnable_pos_emb=False, use_flash_attn=True, use_checkpoint=False, tubelet_size=2, num_classes=0,
                 in_chans=0):
        super().__init__()
        self.encoder = PretrainVisionTransformerEncoder(
            img_size=img_size, patch_size=patch_size, in_chans=encoder_in_chans, num_classes=encoder_num_classes,
            embed_dim=encoder_embed_dim, depth=encoder_depth, num_heads=encoder_num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
            init_values=init_values, tubelet_size=tubelet_size, use_checkpoint=use_checkpoint,
            use_learnable_pos_emb=use_learnable_pos_emb, use_flash_attn=use_flash_attn)
        self.decoder = PretrainVisionTransformerDecoder(
            patch_size=patch_size, num_patches=self.encoder.patch_embed.num_patches, num_classes=decoder_num_classes,
            embed_dim=decoder_embed_dim, depth=decoder_depth, num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
            init_values=init_values, tubelet_size=tubelet_size, use_checkpoint=use_checkpoint, use_flash_attn=use_flash_attn)
        self.encoder_to_decoder = nn.Linear(encoder_embed_dim, decoder_embed_dim, bias=False)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.pos_embed = get_sinusoid_encoding_table(self.encoder.patch_embed.num_patches, decoder_embed_dim)
        trunc_normal_(self.mask_token, std=.02)
        self._pos_dev = None

    def get_num_layers(self):
        return len(self.blocks)  # (raises AttributeError exactly as the reference's does, modeling_pretrain.py:262-263)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'mask_token'}

    def _pos_on(self, device):
        if self._pos_dev is None or self._pos_dev.device != device:
            self._pos_dev = self.pos_embed[0].to(device=device, dtype=torch.float32).contiguous()
        return self._pos_dev

    def forward(self, x, mask, num_masked=None):
        if not x.is_cuda:
            raise TadError(f"PretrainVisionTransformer: input is on {x.device}; the MI355X path runs HIP kernels only (no CPU fallback)")
        vis_tok, mask_tok = token_indices(mask.to(x.device), num_masked)
        x_vis = self.encoder(x, mask, vis_tok=vis_tok)                                          # [B, N_vis, C_e]
        x_vis = ops.LinearFn.apply(x_vis, self.encoder_to_decoder.weight, None)               # [B, N_vis, C_d]
        # the visible tokens keep their (shuffled) order; the positional table is gathered accordingly (:281-287)
        x_full = ops.MaeAssembleFn.apply(x_vis, self.mask_token, self._pos_on(x.device), vis_tok.reshape(-1), mask_tok.reshape(-1))
        return self.decoder(x_full, mask_tok.shape[1])                                          # [B, N_mask, 3*tub*p*p]


def _pretrain(pretrained, enc_dim, enc_depth, enc_heads, dec_dim, dec_heads, **kwargs):
    model = PretrainVisionTransformer(img_size=224, patch_size=16, encoder_embed_dim=enc_dim, encoder_depth=enc_depth,
                                      encoder_num_heads=enc_heads, encoder_num_classes=0, decoder_num_classes=1536,
                                      decoder_embed_dim=dec_dim, decoder_num_heads=dec_heads, mlp_ratio=4, qkv_bias=True,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                      **{k: v for k, v in kwargs.items() if k != "init_ckpt"})
    model.default_cfg = _cfg()
    if pretrained:
        ckpt = torch.load(kwargs["init_ckpt"], map_location="cpu")
        model.load_state_dict(ckpt["model"])
    return model


@register_model
def pretrain_videomae_small_patch16_224(pretrained=False, **kwargs):
    """modeling_pretrain.py:293-315"""
    return _pretrain(pretrained, 384, 12, 6, 192, 3, **kwargs)


@register_model
def pretrain_videomae_base_patch16_224(pretrained=False, **kwargs):
    """modeling_pretrain.py:317-338"""
    return _pretrain(pretrained, 768, 12, 12, 384, 6, **kwargs)


@register_model
def pretrain_videomae_large_patch16_224(pretrained=False, **kwargs):
    """modeling_pretrain.py:340-362"""
    return _pretrain(pretrained, 1024, 24, 16, 512, 8, **kwargs)


@register_model
def pretrain_videomae_huge_patch16_224(pretrained=False, **kwargs):
    """modeling_pretrain.py:364-386 (head_dim 80: attention through the generic f32 kernels)"""
    return _pretrain(pretrained, 1280, 32, 16, 640, 8, **kwargs)
