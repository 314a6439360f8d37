"""Host-side logic of the boundary module (no GPU): attribute tree, state-dict contract, seeded
init identical to the reference's, registry behaviour, and the no-CPU-fallback rule."""
import hashlib

import numpy as np
import pytest
import torch

import golden_recipe as R
import simple_tad_amd as T
from simple_tad_amd._lib import TadError
from oracle import vit_oracle as O


def test_state_dict_contract_matches_reference_keys():
    c = R.TINY
    m = T.VisionTransformer(img_size=c["img_size"], patch_size=c["patch_size"], embed_dim=c["embed_dim"], depth=c["depth"],
                            num_heads=c["num_heads"], mlp_ratio=4, qkv_bias=True, all_frames=c["all_frames"],
                            tubelet_size=c["tubelet_size"], num_classes=c["num_classes"])
    shapes = R.vit_param_shapes(c["embed_dim"], c["depth"], c["num_classes"], tubelet=c["tubelet_size"], patch=c["patch_size"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())          # same keys, same registration order
    assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    assert "pos_embed" not in sd                            # sinusoid table is not a parameter/buffer
    assert m.no_weight_decay() == {'pos_embed', 'cls_token'} and m.get_num_layers() == c["depth"]
    assert m.patch_embed.num_patches == 8 and m.patch_embed.patch_size == (8, 8) and m.patch_embed.tubelet_size == 2
    assert m.num_heads == 2 and hasattr(m.blocks[0].attn, "q_bias") and m.blocks[0].attn.qkv.bias is None


@pytest.mark.parametrize("tag,name,frames,nparams", [("s8", "vit_small_patch16_224", 8, 21880706),
                                                     ("b16", "vit_base_patch16_224", 16, 86228738)])
def test_seeded_init_reproduces_reference_weights(golden, tag, name, frames, nparams):
    """Same construction order + init sequence => same RNG stream => identical weights (checksums from the reference)."""
    g = golden("g4_real_shape")
    torch.manual_seed(0)
    m = T.create_model(name, pretrained=False, num_classes=2, all_frames=frames, tubelet_size=2, final_reduction="fc_norm",
                       use_flash_attn=False, init_scale=1.0, drop_path_rate=0.0, drop_block_rate=None)
    assert sum(p.numel() for p in m.parameters()) == nparams == int(g[f"{tag}.nparams"])
    keys = [str(k) for k in g[f"{tag}.keys"]]
    sd = m.state_dict()
    assert list(sd.keys()) == keys
    gen = torch.Generator().manual_seed(1234)   # the fixture re-randomises 1-D params in order (tools/make_goldens.py g4)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02 + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
    ws = np.array([sd[k].double().sum().item() for k in keys])
    wa = np.array([sd[k].double().abs().sum().item() for k in keys])
    assert np.allclose(ws, g[f"{tag}.wsum"], rtol=0, atol=1e-9) and np.allclose(wa, g[f"{tag}.wabs"], rtol=1e-12, atol=0)


def test_sinusoid_table_bit_exact(golden):
    g = golden("g1_bookkeeping")
    for (n, d) in [(784, 384), (1568, 768)]:
        t = T.get_sinusoid_encoding_table(n, d)
        sha = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        assert (sha == g[f"sha256_{n}_{d}"]).all()


def test_registry_and_legacy_kwargs():
    assert {"vit_small_patch16_224", "vit_base_patch16_224", "vit_large_patch16_224", "vit_huge_patch16_224",
            "vit_base_patch16_384", "vit_large_patch16_384", "vit_large_patch16_512"} <= set(T.list_models())
    m = T.create_model("vit_small_patch16_224", num_classes=2, all_frames=8, drop_block_rate=None, use_mean_pooling=True,
                       drop_path_rate=0.1, init_scale=0.001)
    assert m.final_reduction == "fc_norm" and m.fc_norm is not None and isinstance(m.norm, torch.nn.Identity)
    assert isinstance(m.blocks[0].drop_path, torch.nn.Identity) and abs(m.blocks[-1].drop_path.drop_prob - 0.1) < 1e-7
    assert m.blocks[0].norm1.eps == 1e-6
    m2 = T.create_model("vit_small_patch16_224", num_classes=2, all_frames=8, use_mean_pooling=False)
    assert m2.final_reduction == "cls" and m2.fc_norm is None
    with pytest.raises(RuntimeError):
        T.create_model("no_such_model")


def test_no_cpu_fallback():
    m = T.VisionTransformer(img_size=16, patch_size=8, embed_dim=128, depth=1, num_heads=2, qkv_bias=True, all_frames=4, num_classes=2)
    with pytest.raises(TadError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 4, 16, 16))
    with pytest.raises(TadError):
        m.blocks[0](torch.zeros(1, 8, 128))
    with pytest.raises(AssertionError):  # reference's input-size assert (modeling_finetune.py:188)
        m.patch_embed(torch.zeros(1, 3, 4, 32, 32))


def test_droppath_matches_timm_formula():
    dp = T.DropPath(0.25)
    dp.train()
    dp.forced_mask = torch.tensor([1.0, 0.0, 1.0])
    x = torch.ones(3, 2, 2)
    y = dp(x)
    ref = O.drop_path(x, dp.forced_mask, 0.75)
    assert torch.equal(y, ref)
    dp.eval()
    assert dp(x) is x
