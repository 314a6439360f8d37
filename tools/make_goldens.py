#!/usr/bin/env python3
"""Generate ``tests/golden/*.npz`` by running the REAL reference (imported from
/root/reference, CPU, fp32/fp64) on the deterministic inputs of
``tests/golden_recipe.py``.

Runs only in the build container (the reference never travels to the GPU box).
The fixtures hold data only: inputs are regenerated from the recipe, outputs
are stored.  Nothing from the reference's source text is copied.

Third-party modules the reference imports but this image lacks are satisfied
as follows: ``timm.models.layers.{to_2tuple,trunc_normal_}`` are taken from the
reference's own in-repo copies (run_inference_simple.py:40-105);
``timm.models.layers.drop_path`` is never reached (drop_path_rate=0 in every
fixture); ``register_model`` is the identity decorator; ``flash_attention_class``
is never reached (use_flash_attn=False; it needs the CUDA flash-attn wheel).

usage: python tools/make_goldens.py [--only g3]
"""
import argparse
import hashlib
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden")

import golden_recipe as R  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    cv2 = types.ModuleType("cv2")
    natsort = types.ModuleType("natsort")
    natsort.natsorted = sorted
    sys.modules["cv2"] = cv2
    sys.modules["natsort"] = natsort
    import run_inference_simple as ris

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")

    def _no_drop_path(x, drop_prob=0.0, training=False):
        assert not (training and drop_prob), "fixtures use drop_path_rate=0"
        return x

    layers.drop_path = _no_drop_path
    layers.to_2tuple = ris.to_2tuple
    layers.trunc_normal_ = ris.trunc_normal_
    registry.register_model = lambda f: f
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers,
                        "timm.models.registry": registry})
    fa = types.ModuleType("flash_attention_class")

    class _NoFlash(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("flash-attn is not available; fixtures use use_flash_attn=False")

    fa.FlashAttention = _NoFlash
    sys.modules["flash_attention_class"] = fa
    import modeling_finetune as mf
    import masking_generator as mg
    return ris, mf, mg


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def load_params(module: nn.Module, P):
    sd = module.state_dict()
    assert set(sd.keys()) == set(P.keys()), (sorted(set(sd) ^ set(P)))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    module.load_state_dict({k: v.clone() for k, v in P.items()})


# ----------------------------------------------------------------------------
def g1(mf):
    arrs = {}
    for (n, d) in [(784, 384), (1568, 384), (1568, 768), (1568, 1024), (8, 128)]:
        t = mf.get_sinusoid_encoding_table(n, d)
        assert t.shape == (1, n, d) and t.dtype == torch.float32
        arrs[f"sha256_{n}_{d}"] = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        rows = [r for r in (0, 1, 195, 196, 783, 1567) if r < n]
        arrs[f"rows_{n}_{d}"] = t[0, rows].numpy()
        arrs[f"rowidx_{n}_{d}"] = np.array(rows)
    # patch-index bookkeeping: identity weights, voxel-coded input -> out[b,n,k] = code
    pe = mf.PatchEmbed(img_size=32, patch_size=16, in_chans=3, embed_dim=1536, num_frames=4, tubelet_size=2)
    with torch.no_grad():
        pe.proj.weight.copy_(torch.eye(1536).reshape(1536, 3, 2, 16, 16))
        pe.proj.bias.zero_()
        x = torch.arange(2 * 3 * 4 * 32 * 32, dtype=torch.float32).reshape(2, 3, 4, 32, 32)
        y = pe(x)
    assert y.shape == (2, 8, 1536)
    arrs["patch_codes"] = y.round().to(torch.int32).numpy()
    assert (y - y.round()).abs().max() == 0
    for (img, p, T, tub) in [(224, 16, 8, 2), (224, 16, 16, 2), (32, 16, 4, 2), (16, 8, 4, 2)]:
        arrs[f"num_patches_{img}_{p}_{T}_{tub}"] = np.array(
            mf.PatchEmbed(img_size=img, patch_size=p, embed_dim=8, num_frames=T, tubelet_size=tub).num_patches)
    save("g1_bookkeeping", **arrs)


def g2(mf):
    torch.manual_seed(0)
    arrs = {}
    # PatchEmbed
    pe = mf.PatchEmbed(img_size=32, patch_size=16, in_chans=3, embed_dim=64, num_frames=4, tubelet_size=2).double()
    P = {"proj.weight": R.tensor_for("pe.w", (64, 3, 2, 16, 16), scale=0.02).double(),
         "proj.bias": R.tensor_for("pe.b", (64,), scale=0.02).double()}
    load_params(pe, P)
    x = R.tensor_for("pe.x", (1, 3, 4, 32, 32)).double()
    y = pe(x)
    dy = R.tensor_for("pe.dy", tuple(y.shape)).double()
    y.backward(dy)
    arrs.update({"pe.y": y.detach().numpy(), "pe.dw": pe.proj.weight.grad.numpy(), "pe.db": pe.proj.bias.grad.numpy()})

    # LayerNorm eps=1e-6
    ln = nn.LayerNorm(128, eps=1e-6).double()
    load_params(ln, {"weight": R.tensor_for("ln.w", (128,), scale=0.1, shift=1.0).double(),
                     "bias": R.tensor_for("ln.b", (128,), scale=0.1).double()})
    x = R.tensor_for("ln.x", (3, 50, 128), scale=2.0, shift=0.5).double().requires_grad_()
    y = ln(x)
    dy = R.tensor_for("ln.dy", tuple(y.shape)).double()
    y.backward(dy)
    arrs.update({"ln.y": y.detach().numpy(), "ln.dx": x.grad.numpy(), "ln.dw": ln.weight.grad.numpy(),
                 "ln.db": ln.bias.grad.numpy()})

    # Attention dim=128 heads=2 (d=64), N=100, non-zero q_bias / v_bias
    att = mf.Attention(128, num_heads=2, qkv_bias=True, use_flash_attn=False).double()
    PA = {"q_bias": R.tensor_for("att.qb", (128,), scale=0.1).double(),
          "v_bias": R.tensor_for("att.vb", (128,), scale=0.1).double(),
          "qkv.weight": R.tensor_for("att.qkv", (384, 128), scale=0.08).double(),
          "proj.weight": R.tensor_for("att.pw", (128, 128), scale=0.08).double(),
          "proj.bias": R.tensor_for("att.pb", (128,), scale=0.1).double()}
    load_params(att, PA)
    x = R.tensor_for("att.x", (2, 100, 128)).double().requires_grad_()
    y = att(x)
    dy = R.tensor_for("att.dy", tuple(y.shape)).double()
    y.backward(dy)
    arrs.update({"att.y": y.detach().numpy(), "att.dx": x.grad.numpy()})
    for k, p in att.named_parameters():
        arrs["att.d." + k] = p.grad.numpy()

    # Mlp 128 -> 512 -> 128
    m = mf.Mlp(128, 512).double()
    PM = {"fc1.weight": R.tensor_for("mlp.w1", (512, 128), scale=0.08).double(),
          "fc1.bias": R.tensor_for("mlp.b1", (512,), scale=0.1).double(),
          "fc2.weight": R.tensor_for("mlp.w2", (128, 512), scale=0.08).double(),
          "fc2.bias": R.tensor_for("mlp.b2", (128,), scale=0.1).double()}
    load_params(m, PM)
    x = R.tensor_for("mlp.x", (2, 100, 128)).double().requires_grad_()
    y = m(x)
    dy = R.tensor_for("mlp.dy", tuple(y.shape)).double()
    y.backward(dy)
    arrs.update({"mlp.y": y.detach().numpy(), "mlp.dx": x.grad.numpy()})
    for k, p in m.named_parameters():
        arrs["mlp.d." + k] = p.grad.numpy()

    # Block dim=128 heads=2
    blk = mf.Block(128, 2, mlp_ratio=4., qkv_bias=True, init_values=0., norm_layer=partial(nn.LayerNorm, eps=1e-6),
                   use_flash_attn=False).double()
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    PB = {k: R.tensor_for("blk." + k, s, scale=0.08, shift=1.0 if k.endswith(("norm1.weight", "norm2.weight")) else 0.0).double()
          for k, s in shapes.items()}
    load_params(blk, PB)
    x = R.tensor_for("blk.x", (2, 100, 128)).double().requires_grad_()
    y = blk(x)
    dy = R.tensor_for("blk.dy", tuple(y.shape)).double()
    y.backward(dy)
    arrs.update({"blk.y": y.detach().numpy(), "blk.dx": x.grad.numpy()})
    arrs["blk.keys"] = np.array(list(shapes.keys()))
    for k, p in blk.named_parameters():
        arrs["blk.d." + k] = p.grad.numpy()
    # stored as fp32 (outputs were computed in fp64, so they are correctly-rounded fp32 pins)
    arrs = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in arrs.items()}
    save("g2_ops", **arrs)


def build_tiny(mf, dtype=torch.float64):
    c = R.TINY
    model = mf.VisionTransformer(img_size=c["img_size"], patch_size=c["patch_size"], embed_dim=c["embed_dim"],
                                 depth=c["depth"], num_heads=c["num_heads"], mlp_ratio=4, qkv_bias=True,
                                 norm_layer=partial(nn.LayerNorm, eps=1e-6), all_frames=c["all_frames"],
                                 tubelet_size=c["tubelet_size"], num_classes=c["num_classes"], init_scale=1.0,
                                 drop_path_rate=0.0, use_flash_attn=False)
    shapes = R.vit_param_shapes(c["embed_dim"], c["depth"], c["num_classes"], tubelet=c["tubelet_size"],
                                patch=c["patch_size"])
    assert list(model.state_dict().keys()) == list(shapes.keys()), "state-dict key order differs from recipe"
    P = R.params_for(shapes, seed=3)
    load_params(model, P)
    return model.to(dtype), P


def g3(mf):
    """Tiny full model: fwd, CE loss, all grads, grad-norm, one AdamW step -- fp64 reference run."""
    model, P = build_tiny(mf, torch.float64)
    c = R.TINY
    x = R.tensor_for("tiny.x", (2, 3, c["all_frames"], c["img_size"], c["img_size"]), seed=3).double()
    labels = torch.tensor([0, 1])
    model.train()
    feats = model.forward_features(x)
    logits = model.head(feats)
    loss = nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    # utils.get_grad_norm_ definition
    total = torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads.values()]), 2.0)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999))
    opt.step()
    arrs = {"features": feats.detach().numpy(), "logits": logits.detach().numpy(), "loss": np.array(loss.item()),
            "grad_norm": np.array(total.item()), "keys": np.array(list(grads.keys()))}
    for k, g in grads.items():
        for kk, v in R.summarize(g.float()).items():
            arrs[f"grad.{k}.{kk}"] = v
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.detach().float()).items():
            arrs[f"after.{k}.{kk}"] = v
    # second view: fp32 run of the same thing (what the reference computes by default)
    model32, _ = build_tiny(mf, torch.float32)
    with torch.no_grad():
        arrs["logits_fp32"] = model32(x.float()).numpy()
    arrs = {k: (v.astype(np.float32) if getattr(v, "dtype", None) == np.float64 and v.ndim > 0 else v)
            for k, v in arrs.items()}
    save("g3_tiny_model", **arrs)


def g4(mf, ris):
    """Real-shape logits: S8 B=2 (BASELINE config 1, via run_inference_simple's model) and B16 B=2.
    Weights come from the reference's own seeded init (torch.manual_seed(0), init_scale=1.0);
    per-tensor checksums let the tests prove a regenerated model has identical weights."""
    arrs = {}
    for tag, factory, frames in (("s8", mf.vit_small_patch16_224, 8), ("b16", mf.vit_base_patch16_224, 16)):
        torch.manual_seed(0)
        model = factory(num_classes=2, all_frames=frames, tubelet_size=2, final_reduction="fc_norm",
                        use_flash_attn=False, init_scale=1.0, drop_path_rate=0.0)
        model.eval()
        # exercise biases too: re-randomise every 1-D parameter ~ N(0,0.02) (+1 for norm weights), in order
        g = torch.Generator().manual_seed(1234)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if p.dim() == 1:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02 + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
        torch.manual_seed(1)
        x = torch.randn(2, 3, frames, 224, 224)
        with torch.no_grad():
            feats = model.forward_features(x)
            logits = model.head(feats)
        arrs[f"{tag}.features"] = feats.numpy()
        arrs[f"{tag}.logits"] = logits.numpy()
        arrs[f"{tag}.nparams"] = np.array(sum(p.numel() for p in model.parameters()))
        keys = list(model.state_dict().keys())
        arrs[f"{tag}.keys"] = np.array(keys)
        arrs[f"{tag}.wsum"] = np.array([model.state_dict()[k].double().sum().item() for k in keys])
        arrs[f"{tag}.wabs"] = np.array([model.state_dict()[k].double().abs().sum().item() for k in keys])
        if tag == "s8":
            # the same weights through run_inference_simple's self-contained model (BASELINE config 1);
            # it bakes a softmax into forward (run_inference_simple.py:378-382)
            m2 = ris.VisionTransformerInfer(img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6,
                                            mlp_ratio=4, qkv_bias=True, num_classes=2, all_frames=frames,
                                            tubelet_size=2, final_reduction="fc_norm", use_flash_attn=False,
                                            norm_layer=partial(nn.LayerNorm, eps=1e-6)) \
                if hasattr(ris, "VisionTransformerInfer") else None
            if m2 is not None:
                missing = m2.load_state_dict(model.state_dict(), strict=False)
                m2.eval()
                with torch.no_grad():
                    arrs["s8.infer_probs"] = m2(x).numpy()
                arrs["s8.infer_missing"] = np.array([str(missing)])
    save("g4_real_shape", **arrs)


def g5_stubs():
    """third-party modules the reference's utils.py / optim_factory.py import at module level and this image lacks (tensorboardX,
    timm.utils, the timm optimizer zoo): empty stand-ins, none of them is on a path the fixtures exercise"""
    sys.path.insert(0, REF)
    for name in ("tensorboardX",):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.SummaryWriter = object
            sys.modules[name] = m
    timm_utils = types.ModuleType("timm.utils")
    timm_utils.get_state_dict = lambda *a, **k: None
    timm_utils.accuracy = None
    timm_utils.ModelEma = object
    sys.modules["timm.utils"] = timm_utils
    for sub in ("adafactor", "adahessian", "adamp", "lookahead", "nadam", "novograd", "nvnovograd", "radam",
                "rmsprop_tf", "sgdp"):
        m = types.ModuleType("timm.optim." + sub)
        for cls in ("Adafactor", "Adahessian", "AdamP", "Lookahead", "Nadam", "NovoGrad", "NvNovoGrad", "RAdam",
                    "RMSpropTF", "SGDP"):
            setattr(m, cls, object)
        sys.modules["timm.optim." + sub] = m
    sys.modules.setdefault("timm.optim", types.ModuleType("timm.optim"))


def g5():
    sys.path.insert(0, REF)
    # utils.py imports tensorboardX etc.; restate-free approach: exec only the function's module deps are heavy,
    # so import optim_factory/utils pieces via importlib with stubs
    for name in ("tensorboardX",):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.SummaryWriter = object
            sys.modules[name] = m
    timm_utils = types.ModuleType("timm.utils")
    timm_utils.get_state_dict = lambda *a, **k: None
    timm_utils.accuracy = None
    timm_utils.ModelEma = object
    sys.modules["timm.utils"] = timm_utils
    try:
        import utils as ref_utils
    except Exception as e:  # pragma: no cover
        print("could not import reference utils:", e)
        raise
    sched = ref_utils.cosine_scheduler(1e-3, 1e-6, epochs=3, niter_per_ep=10, warmup_epochs=1)
    sched2 = ref_utils.cosine_scheduler(5e-4, 1e-6, epochs=2, niter_per_ep=7, warmup_epochs=0)
    sched3 = ref_utils.cosine_scheduler(1e-3, 1e-5, epochs=4, niter_per_ep=5, warmup_epochs=1, start_warmup_value=1e-6,
                                        warmup_steps=3)
    g = [torch.full((3, 4), 0.5), torch.arange(5, dtype=torch.float32)]
    params = [nn.Parameter(torch.zeros_like(t)) for t in g]
    for p, t in zip(params, g):
        p.grad = t
    gn = ref_utils.get_grad_norm_(params).item()
    arrs = {"cos_1e-3_1e-6_3_10_1": sched, "cos_5e-4_1e-6_2_7_0": sched2, "cos_warmup_steps": sched3,
            "grad_norm_known": np.array(gn)}
    # layer-decay map: optim_factory imports a timm optimizer zoo -> stub those modules
    for sub in ("adafactor", "adahessian", "adamp", "lookahead", "nadam", "novograd", "nvnovograd", "radam",
                "rmsprop_tf", "sgdp"):
        m = types.ModuleType("timm.optim." + sub)
        for cls in ("Adafactor", "Adahessian", "AdamP", "Lookahead", "Nadam", "NovoGrad", "NvNovoGrad", "RAdam",
                    "RMSpropTF", "SGDP"):
            setattr(m, cls, object)
        sys.modules["timm.optim." + sub] = m
    sys.modules["timm.optim"] = types.ModuleType("timm.optim")
    import optim_factory as of
    names = list(R.vit_param_shapes(128, 12, 2).keys()) + ["pos_embed", "cls_token", "mask_token", "rel_pos_bias.x"]
    arrs["layer_names"] = np.array(names)
    arrs["layer_ids"] = np.array([of.get_num_layer_for_vit(n, 14) for n in names])
    num_layers = 12
    ld = 0.75
    arrs["layer_scales_0.75_12"] = np.array([ld ** (num_layers + 1 - i) for i in range(num_layers + 2)])
    save("g5_schedules", **arrs)


def g6(mg):
    arrs = {}
    np.random.seed(0)
    gen = mg.TubeMaskingGenerator((8, 14, 14), 0.75)
    m = gen()
    arrs["mask_8_14_14_075"] = m.astype(np.uint8)
    arrs["total_masks"] = np.array(gen.total_masks)
    arrs["per_frame"] = np.array(gen.num_masks_per_frame)
    gen2 = mg.TubeMaskingGenerator((8, 14, 14), 0.9)
    arrs["per_frame_09"] = np.array(gen2.num_masks_per_frame)
    save("g6_tube_mask", **arrs)


def g7(ris):
    """Input stage: the reference's per-frame normalisation, run on seeded uint8 frames.
    * run_inference_simple.prepare_image (== run_inference.py:15-34): its single OpenCV call, cv2.cvtColor(img, COLOR_BGR2RGB),
      is served by a channel-reversal stand-in (OpenCV is not installed); everything else is the reference's own torch code.
    * tensor_normalize (ssv2.py:346-362, same text in dota.py:443-460 / kinetics.py / dada.py): imported from ssv2.py with
      MagicMock modules for the loaders' third-party imports (torchvision, decord), which the function does not touch."""
    import unittest.mock as mock
    cv2 = sys.modules["cv2"]
    cv2.COLOR_BGR2RGB = 4
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
    arrs = {}
    frame = R.uint8_for("g7.frame", (32, 48, 3))            # one BGR frame, HWC
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    arrs["prepare_image"] = ris.prepare_image(frame.numpy().copy(), mean, std).numpy()   # [3,H,W] f32, RGB
    arrs["prepare_image_center"] = ris.prepare_image(frame.numpy().copy(), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)).numpy()
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "decord", "PIL", "PIL.Image",
                 "video_transforms", "volume_transforms", "random_erasing", "rand_augment"):
        sys.modules.setdefault(name, mock.MagicMock())   # (the last four are the reference's augmentation files: they need torchvision/PIL)
    import ssv2
    clip = R.uint8_for("g7.clip", (4, 16, 16, 3))            # [T,H,W,C] RGB, the loaders' buffer layout
    arrs["tensor_normalize"] = ssv2.tensor_normalize(clip.clone(), list(mean), list(std)).numpy()   # [T,H,W,C] f32
    # all 256 byte values through both paths (the complete truth table of the arithmetic, per channel)
    ramp = torch.arange(256, dtype=torch.uint8).view(1, 256, 1).repeat(1, 1, 3)   # [1,256,3]
    arrs["ramp_prepare_image"] = ris.prepare_image(ramp.numpy().copy(), mean, std).numpy()
    arrs["ramp_tensor_normalize"] = ssv2.tensor_normalize(ramp.view(1, 1, 256, 3).clone(), list(mean), list(std)).numpy()
    save("g7_input_stage", **arrs)


def _pretrain_engine(mf):
    """The reference's pre-training engine importable on CPU: module-level imports that this image lacks (utils -> tensorboardX...,
    timm.data.constants) are satisfied by MagicMock / the published ImageNet constants.  Returns (modeling_pretrain,
    engine_for_pretraining, run) where run(model, x, mask, normlize_target) drives the REAL train_one_epoch for one step with
    ``nn.MSELoss`` wrapped to record the (outputs, labels) pair the engine builds; ``torch.cuda.*`` calls are no-ops on CPU."""
    import contextlib
    import unittest.mock as mock
    tdc = types.ModuleType("timm.data.constants")
    tdc.IMAGENET_DEFAULT_MEAN, tdc.IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # timm's published constants
    tdata = types.ModuleType("timm.data")
    tdata.constants = tdc
    sys.modules.update({"timm.data": tdata, "timm.data.constants": tdc})
    um = mock.MagicMock()

    class _Logger:
        def __init__(self, **kw):
            pass

        def add_meter(self, *a, **k):
            pass

        def log_every(self, it, *a, **k):
            yield from it

        def update(self, **k):
            pass

        def synchronize_between_processes(self):
            pass

        meters = {}

    um.MetricLogger = _Logger
    # diagnostics only (per-head gradient norms for plots); the engine asserts that they are non-zero after the loop
    um.collect_grad_norms_pretrain = lambda *a, **k: (np.ones((12, 6, 5)), np.ones((12, 6)), np.ones((2,)))
    # The mock stands in for `utils` only while the two reference modules below bind their module-level imports; afterwards the
    # previous entry (the REAL utils when G12 / G13 imported it first, or nothing) is put back, so that a later
    # `import utils as ref_utils` in the same process gets the real module and a full run reaches G12 / G13.
    prev_utils = sys.modules.get("utils")
    sys.modules["utils"] = um
    try:
        import modeling_pretrain as mp
        import engine_for_pretraining as efp
    finally:
        if prev_utils is None:
            sys.modules.pop("utils", None)
        else:
            sys.modules["utils"] = prev_utils

    def run(model, x, mask, normlize_target=True):
        captured = {}

        class _MSE(nn.MSELoss):
            def forward(self, input, target):
                captured["outputs"], captured["labels"] = input.detach().clone(), target.detach().clone()
                return super().forward(input, target)

        def scaler(loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
            loss.backward()
            captured["loss"] = loss.detach().clone()
            return 0.0

        scaler.state_dict = lambda: {"scale": 1.0}
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        with mock.patch.object(efp.nn, "MSELoss", _MSE), mock.patch("torch.cuda.empty_cache"), mock.patch("torch.cuda.synchronize"), \
                mock.patch("torch.cuda.amp.autocast", lambda *a, **k: contextlib.nullcontext()):
            efp.train_one_epoch(model, [(x, mask)], opt, torch.device("cpu"), 0, scaler, max_norm=0, patch_size=16,
                                normlize_target=normlize_target, start_steps=0)
        return captured

    return mp, efp, run


def g8(mf):
    """MAE pre-training path: the real modeling_pretrain.PretrainVisionTransformer (tiny, fp64) driven by the real
    engine_for_pretraining.train_one_epoch for ONE step on CPU (see _pretrain_engine)."""
    mp, efp, run = _pretrain_engine(mf)
    torch.manual_seed(0)
    model = mp.PretrainVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2,
                                         decoder_num_classes=1536, decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=1,
                                         mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=0.,
                                         use_flash_attn=False, tubelet_size=2).double()
    P = R.params_for({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=8)
    model.load_state_dict({k: v.double() for k, v in P.items()})
    x = R.tensor_for("g8.x", (2, 3, 16, 32, 32), seed=8).double()     # PatchEmbed's default num_frames=16 -> 32 tokens
    rng = np.random.RandomState(8)
    per = np.hstack([np.zeros(1), np.ones(3)])                         # tube mask 0.75 on a 2x2 grid, 8 temporal slots
    masks = []
    for _ in range(2):
        rng.shuffle(per)
        masks.append(np.tile(per, (8, 1)).flatten())
    mask = torch.from_numpy(np.stack(masks)).bool()
    captured = run(model, x, mask, True)
    arrs = {"mask": mask.numpy(), "outputs": captured["outputs"].numpy(), "labels": captured["labels"].numpy(),
            "loss": np.array(captured["loss"].item()), "keys": np.array(list(P.keys()))}
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.grad.float()).items():
            arrs[f"grad.{k}.{kk}"] = v
    # un-normalised (normlize_target=False) labels as well
    arrs["labels_raw"] = run(model, x, mask, False)["labels"].numpy()
    save("g8_pretrain", **arrs)


def g10(mf, mg):
    """BASELINE configs[4] at its real size: pretrain_videomae_large_patch16_224 (ViT-L/16 encoder, 12-block decoder as in
    jobs/dapt/pretrain_capdata_large.sh:34-36), 16x224x224, B = 2, tube mask 0.75 (392 visible / 1176 masked tokens per clip),
    driven by the REAL engine_for_pretraining.train_one_epoch for one step in fp64.  The weights (340 M parameters) are not stored:
    they are the reference's own seeded init (torch.manual_seed(0)) with every 1-D parameter re-randomised; per-tensor checksums let
    the tests prove that the regenerated model is identical before outputs are compared."""
    mp, efp, run = _pretrain_engine(mf)
    torch.manual_seed(0)
    model = mp.pretrain_videomae_large_patch16_224(pretrained=False, drop_path_rate=0.0, decoder_depth=12, use_checkpoint=False,
                                                   use_flash_attn=False)
    R.rerandomize_1d(model)
    x = R.clip_for("g10.x", (2, 3, 16, 224, 224))
    mask = R.tube_masks("g10", 2, (8, 14, 14), 0.75, generator_cls=mg.TubeMaskingGenerator)
    keys = list(model.state_dict().keys())
    arrs = {"mask": np.packbits(mask.numpy(), axis=1), "keys": np.array(keys),
            "wsum": np.array([model.state_dict()[k].double().sum().item() for k in keys]),
            "wabs": np.array([model.state_dict()[k].double().abs().sum().item() for k in keys]),
            "nparams": np.array(sum(p.numel() for p in model.parameters()))}
    model = model.double()
    cap = run(model, x.double(), mask, True)
    out, lab = cap["outputs"], cap["labels"]
    assert out.shape == (2, 1176, 1536) and lab.shape == out.shape
    arrs["loss"] = np.array(cap["loss"].item())
    for nm, t in (("outputs", out), ("labels", lab)):
        for kk, v in R.summarize(t.float(), head=4096).items():
            arrs[f"{nm}.{kk}"] = v
        arrs[f"{nm}.rows"] = t[:, R.G10_ROWS].float().numpy()       # a few whole prediction rows per clip
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.grad.float()).items():
            arrs[f"grad.{k}.{kk}"] = v
    arrs["grad_norm"] = np.array(torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in model.parameters()]), 2.0).item())
    save("g10_vitl_mae", **arrs)


def g11(mf):
    """Gradient side of BASELINE configs[2] at its real shape: ViT-B/16 16x224x224, B = 2, forward + CE loss + backward through the
    real reference model (fp64 run of the fp32-valued weights / inputs of G4's b16 case, so the stored numbers are correctly
    rounded): logits, loss, utils.get_grad_norm_ norm and a summary (head + sum + sum of squares) of every gradient."""
    torch.manual_seed(0)
    model = mf.vit_base_patch16_224(num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm", use_flash_attn=False,
                                    init_scale=1.0, drop_path_rate=0.0)
    R.rerandomize_1d(model)
    torch.manual_seed(1)
    x = torch.randn(2, 3, 16, 224, 224)
    labels = torch.tensor([0, 1])
    keys = list(model.state_dict().keys())
    arrs = {"keys": np.array(keys), "wsum": np.array([model.state_dict()[k].double().sum().item() for k in keys])}
    model = model.double().train()
    feats = model.forward_features(x.double())
    logits = model.head(feats)
    loss = nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    arrs.update({"features": feats.detach().float().numpy(), "logits": logits.detach().numpy(), "loss": np.array(loss.item()),
                 "grad_keys": np.array([k for k, _ in model.named_parameters()])})
    arrs["grad_norm"] = np.array(torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in model.parameters()]), 2.0).item())
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.grad.float()).items():
            arrs[f"grad.{k}.{kk}"] = v
    save("g11_vitb_grads", **arrs)


def g12(mf):
    """Fine-tune trajectory (SURVEY 8a rows 12-14 + 8f-1 together): the tiny fp64 model driven by the REAL
    engine_for_finetuning.train_one_epoch for six micro-batches at update_freq 2 = three optimizer steps, with the reference's own
    utils.NativeScalerWithGradNormCount (gradient clipping), utils.cosine_scheduler (lr with one warm-up step, weight decay 0.05 -> 0.1)
    and optim_factory.create_optimizer + LayerDecayValueAssigner (layer decay 0.75).  CPU: GradScaler / autocast disable themselves
    (scale 1, fp64 throughout); torch.cuda.synchronize is patched out."""
    import argparse as _ap
    import unittest.mock as mock
    g5_stubs()
    import utils as ref_utils
    import optim_factory as of
    import engine_for_finetuning as eff
    c = R.G12
    model, P = build_tiny(mf, torch.float64)
    num_layers = model.get_num_layers()
    assigner = of.LayerDecayValueAssigner([c["layer_decay"] ** (num_layers + 1 - i) for i in range(num_layers + 2)])   # run_class_finetuning.py:430-434
    args = _ap.Namespace(opt="adamw", lr=c["base_lr"], weight_decay=c["weight_decay"], opt_eps=1e-8, opt_betas=(0.9, 0.999), momentum=0.9)
    opt = of.create_optimizer(args, model, skip_list=model.no_weight_decay(), get_num_layer=assigner.get_layer_id,
                              get_layer_scale=assigner.get_scale)
    lr_sched = ref_utils.cosine_scheduler(c["base_lr"], c["min_lr"], 1, c["steps"], warmup_epochs=c["warmup_epochs"],
                                          start_warmup_value=c["start_warmup_value"], warmup_steps=c["warmup_steps"])
    wd_sched = ref_utils.cosine_scheduler(c["weight_decay"], c["weight_decay_end"], 1, c["steps"])
    class _Scaler(ref_utils.NativeScalerWithGradNormCount):   # the real scaler; on CPU its GradScaler is disabled and reports no scale
        def state_dict(self):
            d = super().state_dict()
            return d if "scale" in d else {"scale": 1.0}   # (only logged: engine_for_finetuning.py:100)

    scaler = _Scaler()
    losses, norms, lrs, min_lrs = [], [], [], []

    class _Logger(ref_utils.MetricLogger):   # the real logger; update() is tapped to keep every per-step value
        def update(self, **kw):
            if "loss" in kw:
                losses.append(float(kw["loss"]))
            if "grad_norm" in kw:
                norms.append(None if kw["grad_norm"] is None else float(kw["grad_norm"]))
            if "lr" in kw:
                lrs.append(float(kw["lr"]))
            if "min_lr" in kw:
                min_lrs.append(float(kw["min_lr"]))
            super().update(**kw)

    batches = [(x.double(), y, a, b) for x, y, a, b in R.g12_batches()]
    with mock.patch.object(ref_utils, "MetricLogger", _Logger), mock.patch("torch.cuda.synchronize"):
        avg = eff.train_one_epoch(model, nn.CrossEntropyLoss(), batches, opt, torch.device("cpu"), 0, scaler, max_norm=c["clip_grad"],
                                  start_steps=0, lr_schedule_values=lr_sched, wd_schedule_values=wd_sched,
                                  num_training_steps_per_epoch=c["steps"], update_freq=c["update_freq"])
    print("G12 losses", losses, "grad norms", norms, "lr", lrs)
    assert len(losses) == c["micro_batches"] and sum(n is not None for n in norms) == c["steps"]
    arrs = {"loss": np.array(losses), "grad_norm": np.array([np.nan if n is None else n for n in norms]), "lr": np.array(lrs),
            "min_lr": np.array(min_lrs), "lr_schedule": np.asarray(lr_sched), "wd_schedule": np.asarray(wd_sched),
            "avg_keys": np.array(sorted(avg.keys())), "avg_vals": np.array([float(avg[k]) for k in sorted(avg.keys())]),
            "group_lr_scale": np.array([g["lr_scale"] for g in opt.param_groups]),
            "group_weight_decay": np.array([g["weight_decay"] for g in opt.param_groups]),
            "group_size": np.array([len(g["params"]) for g in opt.param_groups]), "keys": np.array(list(P.keys()))}
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.detach().float()).items():
            arrs[f"after.{k}.{kk}"] = v
    save("g12_finetune_trajectory", **arrs)


def g13(mf):
    """MAE pre-training trajectory: the tiny fp64 PretrainVisionTransformer of G8 driven by the REAL engine_for_pretraining.train_one_epoch
    for three steps with the reference's own optimizer factory (optim_factory.create_optimizer(args, model): AdamW, no-decay groups, as
    run_mae_pretraining.py:289-290), utils.NativeScalerWithGradNormCount (clipping) and utils.cosine_scheduler for lr and weight decay."""
    import argparse as _ap
    import contextlib
    import unittest.mock as mock
    g5_stubs()
    import utils as ref_utils            # the real utils first (scaler, scheduler) ...
    import optim_factory as of
    scaler_cls, cosine = ref_utils.NativeScalerWithGradNormCount, ref_utils.cosine_scheduler
    mp, efp, _ = _pretrain_engine(mf)    # ... then the engine, whose `utils` import is the logger mock of _pretrain_engine
    c = R.G13
    torch.manual_seed(0)
    model = mp.PretrainVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=2, encoder_num_heads=2,
                                         decoder_num_classes=1536, decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=1,
                                         mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=0.,
                                         use_flash_attn=False, tubelet_size=2).double()
    P = R.params_for({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=8)
    model.load_state_dict({k: v.double() for k, v in P.items()})
    args = _ap.Namespace(opt="adamw", lr=c["base_lr"], weight_decay=c["weight_decay"], opt_eps=1e-8, opt_betas=c["betas"], momentum=0.9)
    opt = of.create_optimizer(args, model)
    lr_sched = cosine(c["base_lr"], c["min_lr"], 1, c["steps"], warmup_epochs=c["warmup_epochs"], start_warmup_value=c["start_warmup_value"],
                      warmup_steps=c["warmup_steps"])
    wd_sched = cosine(c["weight_decay"], c["weight_decay_end"], 1, c["steps"])

    class _Scaler(scaler_cls):
        def __init__(self):
            super().__init__()
            self.losses, self.norms = [], []

        def __call__(self, loss, optimizer, **kw):
            self.losses.append(float(loss.item()))
            n = super().__call__(loss, optimizer, **kw)
            self.norms.append(float(n))
            return n

        def state_dict(self):
            d = super().state_dict()
            return d if "scale" in d else {"scale": 1.0}

    scaler = _Scaler()
    batches = [(x.double(), m) for x, m in R.g13_batches()]
    with mock.patch("torch.cuda.empty_cache"), mock.patch("torch.cuda.synchronize"), \
            mock.patch("torch.cuda.amp.autocast", lambda *a, **k: contextlib.nullcontext()):
        efp.train_one_epoch(model, batches, opt, torch.device("cpu"), 0, scaler, max_norm=c["clip_grad"], patch_size=16, normlize_target=True,
                            start_steps=0, lr_schedule_values=lr_sched, wd_schedule_values=wd_sched)
    print("G13 losses", scaler.losses, "grad norms", scaler.norms)
    arrs = {"loss": np.array(scaler.losses), "grad_norm": np.array(scaler.norms), "lr_schedule": np.asarray(lr_sched), "wd_schedule": np.asarray(wd_sched),
            "group_weight_decay": np.array([g["weight_decay"] for g in opt.param_groups]), "group_size": np.array([len(g["params"]) for g in opt.param_groups]),
            "group_lr": np.array([g["lr"] for g in opt.param_groups]), "keys": np.array(list(P.keys())),
            "masks": np.stack([m.numpy() for _, m in batches])}
    for k, p in model.named_parameters():
        for kk, v in R.summarize(p.detach()).items():   # fp64: the encoder's updates here are ~1e-6 (gradients far below Adam's eps), under f32 resolution at 1.0
            arrs[f"after.{k}.{kk}"] = v
    save("g13_pretrain_trajectory", **arrs)


def g9():
    """Evaluation metrics: the reference's anaysis/metrics.py (numpy + scikit-learn 1.7.2, both installed here) on
    seeded class-1 probabilities with ties on the threshold grid."""
    sys.path.insert(0, REF)
    from anaysis import metrics as am
    probs, labels = R.eval_probs_labels("g9", 4000)
    (acc, precision, recall, f1, ap, auroc, confmat, pr_curve, roc_curve, mcc, p_t, r_t, acc_t, f1_t) = am.calculate_MORE_metrics(probs, labels)
    arrs = {"at05": np.array([acc, precision, recall, f1]), "ap": np.array(ap), "auroc": np.array(auroc), "confmat": np.array(confmat),
            "mcc": np.array(mcc), "precision_t": np.array(p_t), "recall_t": np.array(r_t), "acc_t": np.array(acc_t), "f1_t": np.array(f1_t),
            "thresholds": np.array(am.THRESHOLDS)}
    a2 = am.calculate_metrics(probs, labels)     # (acc, precision, recall, f1, mAP, auc)
    arrs["calculate_metrics"] = np.array(a2, dtype=np.float64)
    from sklearn.metrics import auc as sk_auc
    arrs["mcc_auc"] = np.array(sk_auc(am.THRESHOLDS, mcc))   # engine_for_frame_finetuning.py:635
    save("g9_eval_metrics", **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.set_num_threads(8)
    ris, mf, mg = import_reference()
    jobs = {"g1": lambda: g1(mf), "g2": lambda: g2(mf), "g3": lambda: g3(mf), "g4": lambda: g4(mf, ris),
            "g5": g5, "g6": lambda: g6(mg), "g7": lambda: g7(ris), "g8": lambda: g8(mf), "g9": g9,
            "g10": lambda: g10(mf, mg), "g11": lambda: g11(mf), "g12": lambda: g12(mf), "g13": lambda: g13(mf)}
    for k, fn in jobs.items():
        if a.only and a.only != k:
            continue
        print("==", k)
        fn()


if __name__ == "__main__":
    main()
