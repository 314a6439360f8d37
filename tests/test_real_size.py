"""Real-size parity pins: BASELINE configs[2] gradients (G11, ViT-B/16 16x224x224, B = 2, fwd + CE loss
+ bwd of the REAL reference model in fp64) and BASELINE configs[4] (G10, ViT-L/16 MAE pre-training step: 392 visible / 1176 masked
tokens per clip, 12-block decoder, driven by the REAL engine_for_pretraining.train_one_epoch in fp64).

The 86 M / 340 M weights are not stored: every model here is built from the same seeded init as the reference's
(torch.manual_seed(0)) and its per-tensor checksums are compared with the fixture's first.

CPU: the oracle against both fixtures (fp32 restatement vs the fp64 reference run).
GPU: * precise mode (split-bf16 Linears, f32 attention) forward AND backward at the real ViT-B shape within north_star's 1e-3;
     * fast (bf16 MFMA) mode with its measured deviation, next to the reference's own bf16-autocast yardstick (3.6e-3 / 4.2e-3);
     * ViT-L/16 MAE step at real size, fast mode;
     * the per-kernel error budget of the fast mode (which operator carries the end-to-end 2.9e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_recipe as R
from oracle import vit_oracle as O


def rell2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def head_err(t, g, key):
    head = torch.from_numpy(g[key + ".head"]).double()
    got = t.detach().double().cpu().flatten()[: head.numel()]
    return ((got - head).norm() / head.norm().clamp_min(1e-30)).item()


def head_err_scaled(t, g, key):
    """error of the stored slice relative to the RMS of the WHOLE tensor (from the stored sum of squares): the first 256 elements of a
    qkv weight gradient are one q-row, whose gradient at random init is two orders of magnitude below the v-rows' -- a per-slice
    relative error then measures bf16 noise against almost nothing"""
    head = torch.from_numpy(g[key + ".head"]).double()
    got = t.detach().double().cpu().flatten()[: head.numel()]
    rms = (float(g[key + ".sqsum"]) / t.numel()) ** 0.5
    return ((got - head).norm() / (head.numel() ** 0.5 * max(rms, 1e-300))).item()


def sq_err(t, g, key):
    sq = float(g[key + ".sqsum"])
    return abs((t.detach().double() ** 2).sum().item() - sq) / max(sq, 1e-300)


def check_weights(model, g):
    sd = model.state_dict()
    keys = [str(k) for k in g["keys"]]
    assert list(sd.keys()) == keys
    ws = np.array([sd[k].double().sum().item() for k in keys])
    wa = np.array([sd[k].double().abs().sum().item() for k in keys])
    # The init draws go through erfinv (trunc_normal_): hosts with different vector units round it differently in the last bit of a few
    # fp32 weights (measured between the build container and the GPU box: per-tensor sums differ by ~5e-9 of the tensor's sum of
    # magnitudes, i.e. ~1e-7 relative per affected weight -- four orders of magnitude below the tolerances tested here)
    assert (np.abs(ws - g["wsum"]) <= 2e-7 * np.maximum(wa, 1e-30)).all(), "regenerated weights differ from the reference's seeded init"
    if "wabs" in g.files:
        assert np.allclose(wa, g["wabs"], rtol=2e-7, atol=0), "regenerated weights differ from the reference's seeded init"


def build_vitb():
    import simple_tad_amd as T
    torch.manual_seed(0)
    m = T.create_model("vit_base_patch16_224", pretrained=False, num_classes=2, all_frames=16, tubelet_size=2, final_reduction="fc_norm",
                       use_flash_attn=False, init_scale=1.0, drop_path_rate=0.0)
    R.rerandomize_1d(m)
    torch.manual_seed(1)
    x = torch.randn(2, 3, 16, 224, 224)
    return m, x, torch.tensor([0, 1])


def build_vitl_mae(golden):
    import simple_tad_amd as T
    import simple_tad_amd.modeling_pretrain  # noqa: F401  (registers the factories)
    g = golden("g10_vitl_mae")
    torch.manual_seed(0)
    m = T.create_model("pretrain_videomae_large_patch16_224", pretrained=False, drop_path_rate=0.0, decoder_depth=12, use_checkpoint=False,
                       use_flash_attn=False)
    R.rerandomize_1d(m)
    x = R.clip_for("g10.x", (2, 3, 16, 224, 224))
    mask = torch.from_numpy(np.unpackbits(g["mask"], axis=1)[:, :1568]).bool()
    return g, m, x, mask


# ------------------------------------------------------------------------------------------------------------------ CPU: oracle pins
def test_oracle_vitb_fwd_bwd_vs_reference_golden(golden):
    g = golden("g11_vitb_grads")
    m, x, y = build_vitb()
    check_weights(m, g)
    P = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    feats = O.forward_features(x, P, depth=12, num_heads=12, tubelet=2, patch=16)
    logits = F.linear(feats, P["head.weight"], P["head.bias"])
    loss = F.cross_entropy(logits, y)
    loss.backward()
    assert rell2(feats.detach(), g["features"]) < 2e-5 and rell2(logits.detach(), g["logits"]) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    keys = [str(k) for k in g["grad_keys"]]
    gn = O.grad_norm([P[k].grad for k in keys])
    assert abs(gn.item() - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    for k in keys:
        assert head_err(P[k].grad, g, "grad." + k) < 1e-3 and sq_err(P[k].grad, g, "grad." + k) < 1e-3, k


def test_tube_masks_match_reference_generator(golden):
    """the fixture's masks come from the reference's TubeMaskingGenerator; this repo's generator draws the same ones"""
    from simple_tad_amd.masking_generator import TubeMaskingGenerator
    g = golden("g10_vitl_mae")
    want = np.unpackbits(g["mask"], axis=1)[:, :1568].astype(bool)
    got = R.tube_masks("g10", 2, (8, 14, 14), 0.75, generator_cls=TubeMaskingGenerator).numpy()
    assert np.array_equal(got, want) and want.sum(1).tolist() == [1176, 1176]
    assert (want.reshape(2, 8, 196) == want.reshape(2, 8, 196)[:, :1]).all()   # the same 147 patches in every temporal slot


def test_oracle_vitl_mae_step_vs_reference_golden(golden):
    g, m, x, mask = build_vitl_mae(golden)
    check_weights(m, g)
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"])
    P = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    out = O.pretrain_forward(x, mask, P, enc_depth=24, enc_heads=16, dec_depth=12, dec_heads=8, tubelet=2, patch=16)
    labels = O.mae_target(x, mask, tubelet=2, patch=16)
    assert out.shape == (2, 1176, 1536)
    assert rell2(labels[:, R.G10_ROWS], g["labels.rows"]) < 1e-5 and head_err(labels, g, "labels") < 1e-5
    assert rell2(out[:, R.G10_ROWS].detach(), g["outputs.rows"]) < 5e-5 and sq_err(out, g, "outputs") < 1e-4
    loss = F.mse_loss(out, labels)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    worst = 0.0
    for k in [n for n, _ in m.named_parameters()]:
        e = max(head_err(P[k].grad, g, "grad." + k), sq_err(P[k].grad, g, "grad." + k))
        worst = max(worst, e)
        assert e < 2e-4, (k, e)   # fp32 restatement against the fp64 run, 36 blocks deep (measured 1.6e-5)
    print("oracle ViT-L MAE worst gradient deviation", worst)


# ------------------------------------------------------------------------------------------------------------------ GPU
def _vitb_step(m, x, y):
    feats = m.forward_features(x.cuda())
    logits = m.head(feats)
    loss = F.cross_entropy(logits, y.cuda())
    loss.backward()
    return feats, logits, loss


@pytest.mark.gpu
def test_precise_mode_backward_at_real_shape_within_1e3(golden):
    """north_star tolerance on the gradient side, at the benchmark's model shape: loss, every gradient (head + sum of squares) and
    the grad-norm of ViT-B/16 16x224x224 within 1e-3 of the reference (engine_for_finetuning.py:67-95 drives this backward)."""
    import simple_tad_amd as T
    g = golden("g11_vitb_grads")
    m, x, y = build_vitb()
    check_weights(m, g)
    m = m.cuda().train()
    T.set_precision("precise")
    try:
        feats, logits, loss = _vitb_step(m, x, y)
    finally:
        T.set_precision("fast")
    e_f, e_l = rell2(feats, g["features"]), rell2(logits, g["logits"])
    assert e_f < 1e-3 and e_l < 1e-3 and abs(loss.item() - float(g["loss"])) < 1e-4, (e_f, e_l, loss.item())
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert list(grads) == [str(k) for k in g["grad_keys"]]
    gn = O.grad_norm([v.float().cpu() for v in grads.values()])
    assert abs(gn.item() - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    worst = ("", 0.0)
    for k, v in grads.items():
        e = max(head_err(v, g, "grad." + k), sq_err(v, g, "grad." + k))
        if e > worst[1]:
            worst = (k, e)
        assert e < 1e-3, (k, e)
    print(f"precise ViT-B real shape: features {e_f:.2e} logits {e_l:.2e} worst gradient {worst[1]:.2e} ({worst[0]})")


@pytest.mark.gpu
def test_fast_mode_backward_at_real_shape_measured_deviation(golden):
    """The benchmarked (bf16 MFMA) mode against the same fixture.  Its deviation is bounded by bf16 operand rounding (2^-9 per operand),
    as the reference's own torch.autocast(bf16) run is (3.6e-3 features / 4.2e-3 logits): the bounds below are
    ~1.5x what was measured on MI355X and are printed so that the bench line's `fast_mode_deviation` can be checked against them."""
    g = golden("g11_vitb_grads")
    m, x, y = build_vitb()
    m = m.cuda().train()
    feats, logits, loss = _vitb_step(m, x, y)
    e_f, e_l = rell2(feats, g["features"]), rell2(logits, g["logits"])
    assert e_f < 4.2e-3 and e_l < 4.4e-3 and abs(loss.item() - float(g["loss"])) < 8e-4   # measured 2.69e-3 / 2.92e-3 / 4.1e-4
    gn = O.grad_norm([p.grad.float().cpu() for p in m.parameters()])
    assert abs(gn.item() - float(g["grad_norm"])) < 5e-3 * float(g["grad_norm"])
    errs = {k: max(head_err_scaled(p.grad, g, "grad." + k), sq_err(p.grad, g, "grad." + k)) for k, p in m.named_parameters()}
    worst = max(errs, key=errs.get)
    med = float(np.median(list(errs.values())))
    slice_rel = {k: head_err(p.grad, g, "grad." + k) for k, p in m.named_parameters()}
    ws = max(slice_rel, key=slice_rel.get)
    print(f"fast ViT-B real shape: features {e_f:.2e} logits {e_l:.2e} loss {loss.item():.6f} grad (slice error / tensor RMS, sqsum) median "
          f"{med:.2e} worst {errs[worst]:.2e} ({worst}); per-slice relative: median {np.median(list(slice_rel.values())):.2e} worst "
          f"{slice_rel[ws]:.2e} ({ws})")
    assert errs[worst] < 1.2e-2 and med < 8e-3   # measured: worst 7.9e-3 (slice error / tensor RMS), median of max(slice/RMS, sqsum) 5.2e-3


@pytest.mark.gpu
def test_half_mode_forward_and_backward_at_real_shape_within_1e3(golden):
    """set_precision("half") = the reference's own autocast arithmetic (float16 operands, f32 accumulation, loss-scaled backward) on the
    SAME kernels at the SAME speed class as the benchmarked bf16 mode: at the benchmark's model shape the outputs sit 3-5x inside
    north_star's 1e-3, and every one of the 162 gradient tensors is within 1e-3 of the reference in its sum of squares and, for the
    stored slice, relative to the tensor's RMS.  What half operands cannot give is 1e-3 relative to EVERY stored slice itself: the
    median slice is at 4.8e-4, a dozen sit between 1e-3 and 1.3e-2 -- the q rows of late qkv weights, whose gradient at seeded init is
    ~100x below their tensor's RMS and ill-conditioned ~x50 against operand rounding in ANY arithmetic (2^-12 * 50 = 1.3e-2 here,
    2^-17 * 50 = 8.8e-4 in the precise mode, 2^-9 * 50 = 1.1e-1 in bf16), and a few MLP weights just above the line.  Counted and bounded."""
    import simple_tad_amd as T
    g = golden("g11_vitb_grads")
    m, x, y = build_vitb()
    m = m.cuda().train()
    scale = 4096.0
    T.set_precision("half")
    try:
        feats = m.forward_features(x.cuda())
        logits = m.head(feats)
        loss = F.cross_entropy(logits, y.cuda())
        (loss * scale).backward()
    finally:
        T.set_precision("fast")
    e_f, e_l = rell2(feats, g["features"]), rell2(logits, g["logits"])
    assert e_f < 5.4e-4 and e_l < 3.0e-4 and abs(loss.item() - float(g["loss"])) < 1e-4, (e_f, e_l, loss.item())   # measured 3.58e-4 / 1.96e-4 / 3.6e-5
    grads = {k: p.grad / scale for k, p in m.named_parameters()}
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
    gn = O.grad_norm([v.float().cpu() for v in grads.values()])
    assert abs(gn.item() - float(g["grad_norm"])) < 4e-4 * float(g["grad_norm"])                                     # measured 2.3e-4
    rms = {k: head_err_scaled(v, g, "grad." + k) for k, v in grads.items()}
    sq = {k: sq_err(v, g, "grad." + k) for k, v in grads.items()}
    sl = {k: head_err(v, g, "grad." + k) for k, v in grads.items()}
    over = sorted(k for k, v in sl.items() if v > 1e-3)
    print(f"half ViT-B real shape: features {e_f:.2e} logits {e_l:.2e} | slice/RMS median {np.median(list(rms.values())):.2e} worst "
          f"{max(rms.values()):.2e} | sqsum worst {max(sq.values()):.2e} | slice-relative median {np.median(list(sl.values())):.2e} worst "
          f"{max(sl.values()):.2e}, over 1e-3: {over}")
    assert max(rms.values()) < 1e-3 and max(sq.values()) < 1e-3            # measured 9.6e-4 (blocks.9.attn.q_bias) / 6.0e-4
    assert float(np.median(list(sl.values()))) < 7.2e-4                      # measured 4.8e-4
    assert len(over) <= 16, over                                             # measured 12 of 162 (bf16 mode: all 162)
    assert max(sl.values()) < 2e-2                                           # measured 1.27e-2, a q row (bf16 mode: 1.1e-1)


@pytest.mark.gpu
def test_vitl_mae_step_at_real_size_vs_reference_golden(golden):
    """BASELINE configs[4]'s workload shape (ViT-L/16 encoder on 392 visible tokens, 12-block decoder on 1568, tube mask 0.75):
    outputs, reconstruction target, loss and every gradient of one pre-training step against the reference engine's fp64 run."""
    from simple_tad_amd import ops
    from simple_tad_amd.engine_pretrain import reconstruction_target
    g, m, x, mask = build_vitl_mae(golden)
    check_weights(m, g)
    m = m.cuda().train()
    out = m(x.cuda(), mask.cuda())
    labels = reconstruction_target(x.cuda(), mask.cuda())
    assert out.shape == (2, 1176, 1536)
    assert rell2(labels[:, R.G10_ROWS], g["labels.rows"]) < 2e-6 and head_err(labels, g, "labels") < 2e-6
    e_rows, e_sq = rell2(out[:, R.G10_ROWS], g["outputs.rows"]), sq_err(out, g, "outputs")
    loss = ops.MseLossFn.apply(out, labels)
    loss.backward()
    errs = {k: max(head_err_scaled(p.grad, g, "grad." + k), sq_err(p.grad, g, "grad." + k)) for k, p in m.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"ViT-L MAE real size (fast): output rows {e_rows:.2e} sqsum {e_sq:.2e} loss {loss.item():.6f} vs {float(g['loss']):.6f} "
          f"grad median {np.median(list(errs.values())):.2e} worst {errs[worst]:.2e} ({worst})")
    assert e_rows < 1e-2 and e_sq < 1e-2 and abs(loss.item() - float(g["loss"])) < 5e-3 * float(g["loss"])
    assert errs[worst] < 6e-2 and float(np.median(list(errs.values()))) < 1.5e-2


@pytest.mark.gpu
def test_fast_mode_error_budget_per_operator(golden):
    """Which operator carries the fast mode's end-to-end deviation: one ViT-B block at the real shape (B = 2, N = 1568, D = 768) in
    precise mode, then with exactly ONE operator class switched to its bf16-MFMA kernel (inputs / outputs of the others stay f32 /
    split-bf16).  The attention branch's three operators each stay below 1.5e-4 of the block's residual contribution; the two MLP
    Linears carry the deviation (2.9e-3 / 2.3e-3: the bf16 rounding of the 3072-wide hidden activation).  Bounds are 1.5x what was
    measured on MI355X; the printed table is the error budget quoted in DESIGN.md section 4.  (Seeded random-init weights, where the
    softmax is near-uniform and averages operand rounding over 1568 keys: the budget of a trained checkpoint may shift towards the
    attention branch -- none is available offline.)"""
    from simple_tad_amd import kernels as K, ops
    from simple_tad_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL
    m, x, _ = build_vitb()
    blk = m.blocks[5].cuda()
    a, mlp = blk.attn, blk.mlp
    torch.manual_seed(3)
    B, N, D, H = 2, 1568, 768, 12
    x0 = (torch.randn(B * N, D, device="cuda") * 1.5).contiguous()
    eps = blk.norm1.eps

    def ln(v, norm, dt):
        return K.layernorm_fwd(v, norm.weight.detach(), norm.bias.detach(), eps, out_dtype=dt, save_stats=False)[0]

    def lin(v, w, b, fast, epi=EPI_BIAS, residual=None):
        if fast:
            vb = v if v.dtype == torch.bfloat16 else K.cast_bf16(v)
            out = K.linear_fwd(vb, ops.w_bf16(w, True), ops._f32c(b), out_dtype=torch.float32, epilogue=epi, residual=residual)[0]
            return out
        return ops.precise_linear(v.float() if v.dtype != torch.float32 else v, w, b, epi, residual=residual, fresh=True)

    def run(fast):
        """fast: set of operator names computed by the bf16 kernels"""
        xn1 = ln(x0, blk.norm1, torch.float32)
        qkv = lin(xn1, a.qkv.weight, ops._qkv_bias(a.q_bias, a.v_bias), "qkv" in fast)
        if "attn" in fast:
            ao = K.attn_fwd(K.cast_bf16(qkv), B, N, H, a.scale, out_dtype=torch.bfloat16, want_lse=False)[0].float()
        else:
            ao = K.attn_fwd_f32(qkv, B, N, H, a.scale)[0]
        x1 = lin(ao, a.proj.weight, a.proj.bias, "proj" in fast, EPI_BIAS_RESIDUAL, x0)
        xn2 = ln(x1, blk.norm2, torch.float32)
        if "fc1" in fast:
            act = K.linear_fwd(K.cast_bf16(xn2), ops.w_bf16(mlp.fc1.weight, True), ops._f32c(mlp.fc1.bias), out_dtype=torch.bfloat16,
                               epilogue=EPI_BIAS_GELU)[0].float()
        else:
            act = lin(xn2, mlp.fc1.weight, mlp.fc1.bias, False, EPI_BIAS_GELU)
        return lin(act, mlp.fc2.weight, mlp.fc2.bias, "fc2" in fast, EPI_BIAS_RESIDUAL, x1)

    with torch.no_grad():
        ref = run(set())
        # the block's contribution (output minus the residual input) is what accumulates through the depth
        denom = (ref - x0).double().norm()
        budget = {}
        for op in ("qkv", "attn", "proj", "fc1", "fc2"):
            budget[op] = ((run({op}) - ref).double().norm() / denom).item()
        budget["all five"] = ((run({"qkv", "attn", "proj", "fc1", "fc2"}) - ref).double().norm() / denom).item()
    print("fast-mode error budget of one ViT-B block (rel-L2 of the block's residual contribution):",
          ", ".join(f"{k} {v:.2e}" for k, v in budget.items()))
    bound = {"qkv": 1.5e-4, "attn": 1.5e-4, "proj": 1.5e-4, "fc1": 4.4e-3, "fc2": 3.5e-3, "all five": 5.0e-3}
    for op, e in budget.items():
        assert e < bound[op], (op, e, bound[op])
