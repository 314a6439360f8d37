"""Pin the CPU oracle (oracle/vit_oracle.py) against golden vectors produced by
the real reference (tools/make_goldens.py).  CPU only."""
import hashlib

import numpy as np
import torch

import golden_recipe as R
from oracle import vit_oracle as O


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ---------------- G1: integer bookkeeping / constants (bit-exact) ----------------
def test_sinusoid_table_bit_exact(golden):
    g = golden("g1_bookkeeping")
    for (n, d) in [(784, 384), (1568, 384), (1568, 768), (1568, 1024), (8, 128)]:
        t = O.sinusoid_table(n, d)
        assert t.shape == (1, n, d) and t.dtype == torch.float32
        sha = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        assert (sha == g[f"sha256_{n}_{d}"]).all(), (n, d)
        rows = g[f"rowidx_{n}_{d}"]
        assert (t[0, rows].numpy() == g[f"rows_{n}_{d}"]).all()


def test_patch_index_bookkeeping_bit_exact(golden):
    g = golden("g1_bookkeeping")
    x = torch.arange(2 * 3 * 4 * 32 * 32, dtype=torch.float32).reshape(2, 3, 4, 32, 32)
    cols = O.im2col_tubelets(x, 2, 16)
    assert (cols.to(torch.int32).numpy() == g["patch_codes"]).all()
    # explicit index formulas
    codes = g["patch_codes"]
    for (tp, hp, wp, c, kt, kh, kw) in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 0, 2, 1, 15, 3), (0, 1, 1, 1, 0, 7, 15)]:
        n = O.token_index(tp, hp, wp, 2, 2)
        k = O.patch_k_index(c, kt, kh, kw, 2, 16)
        voxel = ((c * 4 + (tp * 2 + kt)) * 32 + (hp * 16 + kh)) * 32 + (wp * 16 + kw)
        assert codes[0, n, k] == voxel
    for (img, p, T, tub) in [(224, 16, 8, 2), (224, 16, 16, 2), (32, 16, 4, 2), (16, 8, 4, 2)]:
        assert O.num_patches(img, p, T, tub) == int(g[f"num_patches_{img}_{p}_{T}_{tub}"])
    assert O.num_patches(224, 16, 16, 2) == 1568 and O.num_patches(224, 16, 8, 2) == 784


# ---------------- G2: per-op fp64 ----------------
def test_patch_embed_op(golden):
    g = golden("g2_ops")
    w = R.tensor_for("pe.w", (64, 3, 2, 16, 16), scale=0.02).double().requires_grad_()
    b = R.tensor_for("pe.b", (64,), scale=0.02).double().requires_grad_()
    x = R.tensor_for("pe.x", (1, 3, 4, 32, 32)).double()
    y = O.patch_embed(x, w, b, 2, 16)
    y.backward(R.tensor_for("pe.dy", tuple(y.shape)).double())
    assert rel(y.detach(), g["pe.y"]) < 1e-6
    assert rel(w.grad, g["pe.dw"]) < 1e-6 and rel(b.grad, g["pe.db"]) < 1e-6


def test_layernorm_op(golden):
    g = golden("g2_ops")
    w = R.tensor_for("ln.w", (128,), scale=0.1, shift=1.0).double().requires_grad_()
    b = R.tensor_for("ln.b", (128,), scale=0.1).double().requires_grad_()
    x = R.tensor_for("ln.x", (3, 50, 128), scale=2.0, shift=0.5).double().requires_grad_()
    y = O.layer_norm(x, w, b, 1e-6)
    y.backward(R.tensor_for("ln.dy", tuple(y.shape)).double())
    for got, key in ((y.detach(), "ln.y"), (x.grad, "ln.dx"), (w.grad, "ln.dw"), (b.grad, "ln.db")):
        assert rel(got, g[key]) < 1e-6, key


def _att_params():
    return {"q_bias": R.tensor_for("att.qb", (128,), scale=0.1), "v_bias": R.tensor_for("att.vb", (128,), scale=0.1),
            "qkv.weight": R.tensor_for("att.qkv", (384, 128), scale=0.08),
            "proj.weight": R.tensor_for("att.pw", (128, 128), scale=0.08),
            "proj.bias": R.tensor_for("att.pb", (128,), scale=0.1)}


def test_attention_op(golden):
    g = golden("g2_ops")
    P = {k: v.double().requires_grad_() for k, v in _att_params().items()}
    x = R.tensor_for("att.x", (2, 100, 128)).double().requires_grad_()
    y = O.attention(x, P, "", 2)
    y.backward(R.tensor_for("att.dy", tuple(y.shape)).double())
    assert rel(y.detach(), g["att.y"]) < 1e-6 and rel(x.grad, g["att.dx"]) < 1e-6
    for k, p in P.items():
        assert rel(p.grad, g["att.d." + k]) < 1e-6, k


def test_mlp_op(golden):
    g = golden("g2_ops")
    P = {"fc1.weight": R.tensor_for("mlp.w1", (512, 128), scale=0.08), "fc1.bias": R.tensor_for("mlp.b1", (512,), scale=0.1),
         "fc2.weight": R.tensor_for("mlp.w2", (128, 512), scale=0.08), "fc2.bias": R.tensor_for("mlp.b2", (128,), scale=0.1)}
    P = {k: v.double().requires_grad_() for k, v in P.items()}
    x = R.tensor_for("mlp.x", (2, 100, 128)).double().requires_grad_()
    y = O.mlp(x, P, "")
    y.backward(R.tensor_for("mlp.dy", tuple(y.shape)).double())
    assert rel(y.detach(), g["mlp.y"]) < 1e-6 and rel(x.grad, g["mlp.dx"]) < 1e-6
    for k, p in P.items():
        assert rel(p.grad, g["mlp.d." + k]) < 1e-6, k


def test_block_op(golden):
    g = golden("g2_ops")
    keys = [str(k) for k in g["blk.keys"]]
    shapes = {"norm1.weight": (128,), "norm1.bias": (128,), "attn.q_bias": (128,), "attn.v_bias": (128,),
              "attn.qkv.weight": (384, 128), "attn.proj.weight": (128, 128), "attn.proj.bias": (128,),
              "norm2.weight": (128,), "norm2.bias": (128,), "mlp.fc1.weight": (512, 128), "mlp.fc1.bias": (512,),
              "mlp.fc2.weight": (128, 512), "mlp.fc2.bias": (128,)}
    assert sorted(keys) == sorted(shapes)
    P = {k: R.tensor_for("blk." + k, shapes[k], scale=0.08,
                         shift=1.0 if k.endswith(("norm1.weight", "norm2.weight")) else 0.0).double().requires_grad_()
         for k in keys}
    x = R.tensor_for("blk.x", (2, 100, 128)).double().requires_grad_()
    y = O.block(x, P, "", 2)
    y.backward(R.tensor_for("blk.dy", tuple(y.shape)).double())
    assert rel(y.detach(), g["blk.y"]) < 1e-6 and rel(x.grad, g["blk.dx"]) < 1e-6
    for k, p in P.items():
        assert rel(p.grad, g["blk.d." + k]) < 1e-6, k


# ---------------- G3: tiny full model, one training step ----------------
def tiny_setup(dtype=torch.float64):
    c = R.TINY
    shapes = R.vit_param_shapes(c["embed_dim"], c["depth"], c["num_classes"], tubelet=c["tubelet_size"], patch=c["patch_size"])
    P = {k: v.to(dtype).requires_grad_() for k, v in R.params_for(shapes, seed=3).items()}
    x = R.tensor_for("tiny.x", (2, 3, c["all_frames"], c["img_size"], c["img_size"]), seed=3).to(dtype)
    kw = dict(depth=c["depth"], num_heads=c["num_heads"], tubelet=c["tubelet_size"], patch=c["patch_size"])
    return P, x, kw


def test_tiny_model_training_step(golden):
    g = golden("g3_tiny_model")
    P, x, kw = tiny_setup()
    feats = O.forward_features(x, P, **kw)
    logits = torch.nn.functional.linear(feats, P["head.weight"], P["head.bias"])
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor([0, 1]))
    loss.backward()
    assert rel(feats.detach(), g["features"]) < 1e-6
    assert rel(logits.detach(), g["logits"]) < 1e-6
    assert abs(loss.item() - float(g["loss"])) < 1e-9
    keys = [str(k) for k in g["keys"]]
    assert keys == list(P.keys())
    gn = O.grad_norm([P[k].grad for k in keys])
    assert abs(gn.item() - float(g["grad_norm"])) < 1e-9 * max(1, float(g["grad_norm"]))
    for k in keys:
        R.check_summary(P[k].grad, g, "grad." + k, rtol=2e-6)
        wd = 0.05
        newp, _, _ = O.adamw_step(P[k].detach(), P[k].grad, torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1, 1e-3, wd)
        R.check_summary(newp, g, "after." + k, rtol=2e-6)
    # the reference's default fp32 run agrees with its fp64 run to fp32 rounding
    assert rel(g["logits_fp32"], g["logits"]) < 1e-5


# ---------------- G5 / G6 ----------------
def test_schedules_and_layer_decay(golden):
    g = golden("g5_schedules")
    assert np.array_equal(O.cosine_scheduler(1e-3, 1e-6, 3, 10, warmup_epochs=1), g["cos_1e-3_1e-6_3_10_1"])
    assert np.array_equal(O.cosine_scheduler(5e-4, 1e-6, 2, 7, warmup_epochs=0), g["cos_5e-4_1e-6_2_7_0"])
    assert np.array_equal(O.cosine_scheduler(1e-3, 1e-5, 4, 5, warmup_epochs=1, start_warmup_value=1e-6, warmup_steps=3),
                          g["cos_warmup_steps"])
    names = [str(n) for n in g["layer_names"]]
    assert [O.layer_id_for_vit(n, 14) for n in names] == list(g["layer_ids"])
    gn = O.grad_norm([torch.full((3, 4), 0.5), torch.arange(5, dtype=torch.float32)])
    assert abs(gn.item() - float(g["grad_norm_known"])) < 1e-6
    assert np.allclose([0.75 ** (13 - i) for i in range(14)], g["layer_scales_0.75_12"], rtol=0, atol=0)


def test_tube_mask(golden):
    g = golden("g6_tube_mask")
    m = O.tube_mask((8, 14, 14), 0.75, np.random.RandomState(0))
    assert m.shape == (8 * 196,) and int(m.sum()) == int(g["total_masks"]) == 8 * 147
    assert int(g["per_frame"]) == 147 and int(g["per_frame_09"]) == 176
    fr = m.reshape(8, 196)
    assert (fr == fr[0]).all()
    # np.random.seed(0) + np.random.shuffle == RandomState(0).shuffle: identical stream
    assert np.array_equal(m.astype(np.uint8), g["mask_8_14_14_075"])
