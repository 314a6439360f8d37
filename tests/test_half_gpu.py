"""IEEE-half operand twins (tad_*_f16) against the fp64 oracle on the SAME half-rounded inputs, through the C ABI.

The half entry points are the bf16 kernels compiled for the other 16-bit format (csrc/common.h), so layouts, schedules and edge
handling are covered by tests/test_kernels_gpu.py; what is format-specific -- the conversions, the MFMA opcode, the 16-bit epilogue
reads (GELU backward, LayerNorm backward) and writes -- is checked here, with the tighter bounds half affords: one half ulp of the
tensor scale is 2^-11 (bf16: 2^-8), and the probabilities P / dS inside attention are rounded to 11 bits instead of 8."""
import math

import pytest
import torch

import golden_recipe as R
from attn_util import prescaled_pair
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-3
F16_ULP = 2.0 ** -11
ATT_TOL_F16 = 6e-4   # measured 2.1e-4 / 3.5e-4 (rel-L2 / max): the bf16 kernels' 1.5e-3 / 3e-3 divided by 8


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from simple_tad_amd import kernels, _lib
    _lib.load()
    return kernels


def dev(t):
    return t.cuda().contiguous()


def hf(t):  # round to half and back
    return t.to(torch.float16).to(torch.float32)


def errs(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def check(a, b, tol=TOL, what=""):
    e1, e2 = errs(a, b)
    assert e1 <= tol and e2 <= tol, f"{what}: max-rel {e1:.3e} l2-rel {e2:.3e} > {tol}"


def test_cast_and_transposes_bit_exact(K):
    x = R.tensor_for("hcast.x", (1237, 77), scale=3.0)
    x[0, :6] = torch.tensor([0.0, -0.0, 1e-7, 65504.0, 70000.0, -1e-9])  # subnormal, max finite, overflow -> inf, underflow -> -0
    y = K.cast_op16(dev(x), dtype=torch.float16)
    assert y.dtype == torch.float16 and torch.equal(y.cpu().view(torch.int16), x.to(torch.float16).view(torch.int16))
    w = R.tensor_for("hcast.w", (304, 136))
    wt = K.transpose_cast_op16(dev(w), dtype=torch.float16)
    assert torch.equal(wt.cpu().view(torch.int16), w.t().contiguous().to(torch.float16).view(torch.int16))
    s = K.scale_cast_op16(dev(w), rowscale=dev(torch.full((304,), 0.5)), rows_per_scale=1, dtype=torch.float16)
    assert torch.equal(s.cpu().view(torch.int16), (w * 0.5).to(torch.float16).view(torch.int16))
    # 16-bit transposes move bit patterns: one entry point for both formats
    src = dev(w).to(torch.float16)
    dst = torch.empty(136 * 304, dtype=torch.float16, device="cuda")
    K.transpose_bf16_batched(src.reshape(-1), dst, K.transpose_table([(0, 304, 136)]).cuda())
    assert torch.equal(dst.view(136, 304).cpu(), src.t().contiguous().cpu())


def test_mixed_formats_are_refused(K):
    x = dev(torch.randn(64, 64)).to(torch.float16)
    w = dev(torch.randn(64, 64)).to(torch.bfloat16)
    from simple_tad_amd._lib import TadError
    with pytest.raises(TadError):
        K.linear_fwd(x, w)
    with pytest.raises(TadError):
        K.linear_fwd(x, x, out_dtype=torch.bfloat16)


@pytest.mark.parametrize("M,N,Kd", [(300, 200, 192), (2500, 768, 256), (25088 + 77, 1024, 128)])
def test_linear_fwd_epilogues(K, M, N, Kd):
    x = hf(R.tensor_for(f"hl.x{M}", (M, Kd)))
    w = hf(R.tensor_for(f"hl.w{M}", (N, Kd), scale=0.05))
    b = R.tensor_for(f"hl.b{M}", (N,), scale=0.1)
    res = R.tensor_for(f"hl.r{M}", (M, N))
    xh, wh = dev(x).half(), dev(w).half()
    ref = x.double() @ w.double().t() + b.double()
    y, _ = K.linear_fwd(xh, wh, dev(b), out_dtype=torch.float32)
    check(y, ref, what="linear f32")
    y16, _ = K.linear_fwd(xh, wh, dev(b))
    assert y16.dtype == torch.float16
    check(y16.float(), ref, tol=F16_ULP, what="linear f16")
    a, h = K.linear_fwd(xh, wh, dev(b), epilogue=K.EPI_BIAS_GELU, want_preact=True)
    check(h.float(), ref, tol=F16_ULP, what="pre-activation")
    check(a.float(), torch.nn.functional.gelu(ref), tol=2 * F16_ULP, what="gelu")
    yr, _ = K.linear_fwd(xh, wh, dev(b), out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=dev(res))
    check(yr, ref + res.double(), what="residual")
    q = K.linear_fwd_qkv(xh[:, :], dev(hf(R.tensor_for(f"hl.wq{M}", (3 * 64, Kd), scale=0.05))).half(), dev(b[:64]), dev(b[64:128]))
    assert q.dtype == torch.float16 and q.shape == (M, 192)


def test_linear_bwd(K):
    M, N, Kd = 3000, 384, 1536
    dy = hf(R.tensor_for("hb.dy", (M, N)))
    w = hf(R.tensor_for("hb.w", (N, Kd), scale=0.05))
    x = hf(R.tensor_for("hb.x", (M, Kd)))
    hpre = hf(R.tensor_for("hb.h", (M, Kd), scale=1.5))
    dyh, xh = dev(dy).half(), dev(x).half()
    wT = dev(w.t().contiguous()).half()
    dx = K.linear_bwd_input(dyh, wT, out_dtype=torch.float32)
    ref = dy.double() @ w.double()
    check(dx, ref, what="dx")
    hh = hpre.double()
    gp = 0.5 * (1 + torch.erf(hh / math.sqrt(2))) + hh * torch.exp(-hh * hh / 2) / math.sqrt(2 * math.pi)
    dxg = K.linear_bwd_input(dyh, wT, gelu_preact=dev(hpre).half())
    check(dxg.float(), ref * gp, tol=2 * F16_ULP, what="dx through GELU'")
    dW, db = K.linear_bwd_weight(dyh, xh)
    check(dW, dy.double().t() @ x.double(), what="dW")
    check(db, dy.double().sum(0), what="db")
    cs = K.colsum_bf16(dyh)
    check(cs, dy.double().sum(0), what="colsum f16")
    # qkv form: bias sums of the first / last third in place
    dWq = torch.zeros(N, Kd, device="cuda")
    dq, dv = torch.zeros(N // 3, device="cuda"), torch.zeros(N // 3, device="cuda")
    K.linear_bwd_weight_qkv(dyh, xh, dWq, dq, dv, accumulate=True)
    check(dWq, dy.double().t() @ x.double(), what="dW qkv")
    check(dq, dy.double().sum(0)[: N // 3], what="dq_bias")
    check(dv, dy.double().sum(0)[2 * N // 3:], what="dv_bias")



def test_half_twins_of_the_round5_linear_plans(K):
    """The IEEE-half instantiations of what round 5 added to the Linear plans: the short-K plan (192 x 128 as four waves / 128 x 128 tiles,
    tad_linear_tuning("short_k")) bit-identical to the planned 256 x 256 launches, and the pair launch of two weight gradients
    (tad_linear_bwd_weight_pair_f16) right against f64 products of the same operands."""
    M = 20000 + 7
    try:
        for N, Kd, mode in ((1536, 384, "gelu"), (384, 384, "res"), (1152, 384, "plain"), (2048, 512, "dgelu")):
            g = torch.Generator().manual_seed(N + Kd)
            x = torch.randn(M, Kd, generator=g).cuda().half()
            w = (torch.randn(N, Kd, generator=g) * 0.05).cuda().half()
            b, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
            h = torch.randn(M, N, generator=g).cuda().half()
            outs = []
            for sk in (0, 1):
                K.linear_tuning(**{**K.LINEAR_TUNING_DEFAULTS, "short_k": sk})
                if mode == "gelu":
                    o = K.linear_fwd(x, w, b, epilogue=K.EPI_BIAS_GELU, want_preact=True)
                elif mode == "res":
                    o = (K.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=K.EPI_BIAS_RESIDUAL, residual=res)[0],)
                elif mode == "dgelu":
                    o = (K.linear_bwd_input(x, w, gelu_preact=h),)
                else:
                    o = (K.linear_fwd(x, w, b)[0],)
                outs.append([t.clone() for t in o])
            assert all(torch.equal(a, c) for a, c in zip(*outs)), f"short-K plan differs ({N}, {Kd}, {mode})"
    finally:
        K.linear_tuning(**K.LINEAR_TUNING_DEFAULTS)
    M, N1, N2, Kd = 25088, 2304, 768, 768
    g = torch.Generator().manual_seed(1)
    dy1, x1 = torch.randn(M, N1, generator=g).cuda().half(), torch.randn(M, Kd, generator=g).cuda().half()
    dy2, x2 = torch.randn(M, N2, generator=g).cuda().half(), torch.randn(M, Kd, generator=g).cuda().half()
    dW1, dW2 = torch.zeros(N1, Kd, device="cuda"), torch.zeros(N2, Kd, device="cuda")
    dq, dv = torch.zeros(N1 // 3, device="cuda"), torch.zeros(N1 // 3, device="cuda")
    K.linear_bwd_weight_pair(dy1, x1, dW1, dq, dv, dy2, x2, dW2, accumulate=False)
    for got, ref in ((dW1[:8], dy1[:, :8].double().t() @ x1.double()), (dW2[-8:], dy2[:, -8:].double().t() @ x2.double()),
                     (dq, dy1[:, :N1 // 3].double().sum(0)), (dv, dy1[:, 2 * (N1 // 3):].double().sum(0))):
        assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("rows,D", [(150, 128), (1030, 768), (9, 1280)])
def test_layernorm(K, rows, D):
    x = R.tensor_for(f"hln.x{D}", (rows, D), scale=2.0, shift=0.5)
    w = R.tensor_for(f"hln.w{D}", (D,), scale=0.1, shift=1.0)
    b = R.tensor_for(f"hln.b{D}", (D,), scale=0.1)
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = O.layer_norm(xd, wd, bd, 1e-6)
    y16, mean, rstd = K.layernorm_fwd(dev(x), dev(w), dev(b), 1e-6, out_dtype=torch.float16)
    assert y16.dtype == torch.float16
    check(y16.float(), ref, tol=F16_ULP, what="ln fwd f16")
    dy = hf(R.tensor_for(f"hln.dy{D}", (rows, D)))
    dres = R.tensor_for(f"hln.dres{D}", (rows, D))
    ref.backward(dy.double())
    dx, dxb, dg, db, cs = K.layernorm_bwd(dev(dy).half(), dev(x), dev(w), mean, rstd, dres=dev(dres), want_bf16=True, want_colsum=True)
    assert dxb.dtype == torch.float16
    check(dx, xd.grad + dres.double(), what="ln dx")
    check(dxb.float(), xd.grad + dres.double(), tol=F16_ULP, what="ln dx f16")
    check(dg, wd.grad, what="ln dgamma")
    check(db, bd.grad, what="ln dbeta")
    check(cs, (xd.grad + dres.double()).sum(0), what="ln colsum", tol=2e-3 if rows > 1000 else TOL)


def _attn_ref(qkv, B, N, H, scale, dout):
    qd = qkv.double().reshape(B, N, -1).requires_grad_()
    ref = O.attention_core(qd, H, scale)
    ref.backward(dout.double().reshape(B, N, -1))
    return ref.detach(), qd.grad


@pytest.mark.parametrize("prescaled", [True, False], ids=["q_prescaled", "plain_q"])
@pytest.mark.parametrize("B,N,H", [(2, 1568, 2), (1, 200, 3), (3, 64, 1)])
def test_attention_fwd_bwd(K, B, N, H, prescaled):
    scale = 0.125
    qkv = hf(R.tensor_for(f"hatt.qkv{N}", (B * N, 3 * H * 64), scale=1.0))
    dout = hf(R.tensor_for(f"hatt.do{N}", (B * N, H * 64)))
    opnd = qkv
    if prescaled:  # the production contract: q third = q * scale * log2e, rounded once (tests/attn_util.py)
        opnd, qkv = prescaled_pair(qkv, B, N, H, scale, hf)
    ref, ref_dqkv = _attn_ref(qkv, B, N, H, scale, dout)
    qh = dev(opnd).half()
    out, lse, lo = K.attn_fwd(qh, B, N, H, scale, want_lo=True, q_prescaled=prescaled)
    assert out.dtype == torch.float16 and lo.dtype == torch.float16
    check(out.float().reshape(B, N, -1), ref, tol=ATT_TOL_F16, what="attn fwd f16")
    # (out + lo reproduces the kernel's f32 accumulator: its distance to the oracle is the rounding of P inside the kernel, not of out)
    check(out.double().reshape(B, N, -1) + lo.double().reshape(B, N, -1), ref, tol=ATT_TOL_F16, what="attn fwd f16 + residual")
    q4 = qkv.double().reshape(B, N, 3, H, 64)
    s = torch.einsum("bnhd,bmhd->bhnm", q4[:, :, 0], q4[:, :, 1]) * scale
    assert (lse.cpu().double() - torch.logsumexp(s, -1)).abs().max().item() < 2e-4
    dqkv = K.attn_bwd(qh, out, dev(dout).half(), lse, B, N, H, scale, out_lo=lo, q_prescaled=prescaled)
    assert dqkv.dtype == torch.float16
    g, r = dqkv.float().cpu().reshape(B, N, 3, H, 64), ref_dqkv.reshape(B, N, 3, H, 64)
    for i, nm in enumerate("qkv"):
        check(g[:, :, i], r[:, :, i], tol=2 * ATT_TOL_F16, what=f"attn d{nm} f16")


def test_exact_delta_rescues_dq_dk_when_values_share_a_common_component(K):
    """delta = rowsum(dO * O) has to cancel against the dP = dO V^T the kernels recompute.  With a large component common to all
    value rows every dP_k is almost delta, dS = P (dP - delta) is a small difference, and a delta taken of the ROUNDED output leaves
    dO . (O - round(O)) behind -- an error proportional to P that lands in dQ and dK in full.  The forward's rounding residual
    (out_lo) removes it; both operand formats."""
    B, N, H, scale = 1, 512, 2, 0.125
    qkv = R.tensor_for("hatt.common", (B * N, 3, H, 64), scale=1.0)
    qkv[:, 2] = 0.05 * qkv[:, 2] + 4.0 * R.tensor_for("hatt.vbar", (1, H, 64))   # V = small variation around a big common row
    qkv[:, 1] = qkv[:, 1] + 2.0 * R.tensor_for("hatt.kbar", (1, H, 64))           # ... and K with a common component as well
    qkv = qkv.reshape(B * N, -1)
    dout = R.tensor_for("hatt.do2", (B * N, H * 64))
    for dt, rnd in ((torch.float16, hf), (torch.bfloat16, lambda t: t.to(torch.bfloat16).float())):
        qr, dr = rnd(qkv), rnd(dout)
        _, ref_dqkv = _attn_ref(qr, B, N, H, scale, dr)
        qd = dev(qr).to(dt)
        out, lse, lo = K.attn_fwd(qd, B, N, H, scale, want_lo=True)
        e = {}
        for name, res in (("rounded", None), ("exact", lo)):
            d = K.attn_bwd(qd, out, dev(dr).to(dt), lse, B, N, H, scale, out_lo=res).float().cpu().reshape(B, N, 3, H, 64)
            r = ref_dqkv.reshape(B, N, 3, H, 64)
            e[name] = [errs(d[:, :, i], r[:, :, i])[1] for i in range(3)]
        print(dt, "rel-L2 of (dq, dk, dv):", e)
        assert e["exact"][0] < 0.25 * e["rounded"][0] and e["exact"][1] < 0.5 * e["rounded"][1]   # dQ, dK
        assert abs(e["exact"][2] - e["rounded"][2]) < 1e-6                                          # dV does not depend on delta


def test_im2col_and_patch_embed(K):
    x = torch.randint(0, 256, (3, 3, 4, 16, 16)).float()  # exact in half
    cols = K.im2col_tubelets(dev(x), 2, 8, dtype=torch.float16)
    assert cols.dtype == torch.float16 and torch.equal(cols.cpu().float().reshape(3, 8, 384), O.im2col_tubelets(x, 2, 8))
    x = R.tensor_for("hpe.x", (1, 3, 4, 32, 32))
    w = R.tensor_for("hpe.w", (64, 3, 2, 16, 16), scale=0.02)
    b = R.tensor_for("hpe.b", (64,), scale=0.02)
    pos = O.sinusoid_table(8, 64)[0]
    out, cols = K.patch_embed_fwd(dev(x), dev(w.reshape(64, -1)).half(), dev(b), dev(pos), 2, 16)
    check(out, O.patch_embed(hf(x).double(), hf(w).double(), b.double(), 2, 16) + pos.double(), what="patch_embed f16")
    assert cols.dtype == torch.float16
    frames = torch.randint(0, 256, (1, 4, 16, 16, 3), dtype=torch.uint8)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    c8 = K.im2col_tubelets_u8(dev(frames), 2, 8, mean, std, dtype=torch.float16)
    xf = ((frames.float() / 255.0 - torch.tensor(mean)) / torch.tensor(std)).permute(0, 4, 1, 2, 3).contiguous()
    assert torch.equal(c8.cpu(), K.im2col_tubelets(dev(xf), 2, 8, dtype=torch.float16).cpu())


def test_adamw_half_mirror_and_skipped_step(K):
    from simple_tad_amd._lib import ADAMW_CHUNK
    n = 3 * ADAMW_CHUNK
    g = torch.Generator().manual_seed(0)
    p = dev(torch.randn(n, generator=g))
    grad = dev(torch.randn(n, generator=g) * 1024.0)  # "loss-scaled" by 1024
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    mirror = torch.zeros(n, dtype=torch.float16, device="cuda")
    cg = torch.zeros(3, dtype=torch.uint8, device="cuda")
    p0 = p.clone()
    ref_p = p.clone().cpu().double().requires_grad_()
    opt = torch.optim.AdamW([ref_p], lr=1e-2, weight_decay=0.05)
    ref_p.grad = (grad.cpu().double() / 1024.0)
    opt.step()
    part = torch.zeros(3, device="cuda")
    # a skipped step (grad_scale == 0: the scaler found an inf) leaves everything untouched but still reports the norm
    K.adamw_step(p, grad, m, v, cg, [1e-2], [0.05], [1], 0.9, 0.999, 1e-8, param_bf16=mirror, grad_scale=torch.zeros(1, device="cuda"),
                 sumsq_partials=part)
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0 and float(mirror.float().abs().max()) == 0.0
    assert abs(float(part.sum()) - float((grad.double() ** 2).sum())) < 1e-5 * float((grad.double() ** 2).sum())
    K.adamw_step(p, grad, m, v, cg, [1e-2], [0.05], [1], 0.9, 0.999, 1e-8, param_bf16=mirror,
                 grad_scale=torch.full((1,), 1.0 / 1024.0, device="cuda"))
    check(p, ref_p.detach(), tol=2e-6, what="adamw with the loss scale removed in the kernel")
    assert torch.equal(mirror.cpu().view(torch.int16), p.cpu().to(torch.float16).view(torch.int16))


def test_half_mode_training_step_follows_the_oracle_with_loss_scaling():
    """set_precision("half") end to end on a small model: engine.NativeScalerWithGradNormCount scales the loss (GradScaler rules), the
    fused AdamW removes the scale; loss, gradient norm and the parameter update follow the fp64 oracle, tighter than the bf16 mode does."""
    import simple_tad_amd as T
    from simple_tad_amd import engine
    import torch.nn.functional as F
    torch.manual_seed(0)
    cfg = dict(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, all_frames=4, tubelet_size=2, num_classes=2)
    x = torch.randn(4, 3, 4, 32, 32)
    y = torch.tensor([0, 1, 1, 0])
    res = {}
    for mode in ("fast", "half"):
        torch.manual_seed(1)
        m = T.VisionTransformer(mlp_ratio=4, qkv_bias=True, init_scale=1.0, **cfg).cuda().train()
        P0 = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
        T.set_precision(mode)
        try:
            opt = engine.create_optimizer(m, lr=1e-3, weight_decay=0.05)
            scaler = engine.NativeScalerWithGradNormCount(m)
            assert scaler.scaling() == (mode == "half") and scaler.state_dict()["scale"] == (65536.0 if mode == "half" else 1.0)
            loss = F.cross_entropy(m(x.cuda()), y.cuda())
            norm = scaler(loss, opt, parameters=list(m.parameters()))
            torch.cuda.synchronize()
        finally:
            T.set_precision("fast")
        P = {k: v.clone().requires_grad_() for k, v in P0.items()}
        ref_loss = F.cross_entropy(O.forward(x.double(), P, depth=2, num_heads=2, tubelet=2, patch=16, eps=m.blocks[0].norm1.eps), y)
        ref_loss.backward()
        ref_norm = O.grad_norm([v.grad for v in P.values() if v.grad is not None])
        res[mode] = (abs(loss.item() - ref_loss.item()), abs(float(norm) - float(ref_norm)) / float(ref_norm))
        assert scaler.skipped_steps == 0 and math.isfinite(float(norm))
        upd = sum(float((v.detach().double().cpu() - P0[k]).abs().sum()) for k, v in m.state_dict().items())
        assert upd > 0.0
    print("tiny model, |loss - oracle| and grad-norm deviation: ", res)
    assert res["half"][0] < 2e-4 and res["half"][1] < 1e-3
    assert res["half"][1] < res["fast"][1]


def test_head_dim_80_model_in_half_mode_follows_the_oracle_and_its_dropout_is_reproducible():
    """The "huge" geometry (head_dim 80, modeling_finetune.py:390-398) through the f16 twins of the 16-bit attention kernels: loss and
    gradients of a small model against the fp64 oracle inside the half-mode band, and attn_drop > 0 (same kernels, keep mask from the
    torch RNG seed) reproducible in training and absent in eval."""
    import simple_tad_amd as T
    import torch.nn.functional as F
    cfg = dict(img_size=32, patch_size=16, embed_dim=320, depth=2, num_heads=4, mlp_ratio=4, qkv_bias=True, all_frames=4, tubelet_size=2,
               num_classes=2, init_scale=1.0)
    torch.manual_seed(3)
    m = T.VisionTransformer(**cfg).cuda().train()
    assert m.blocks[0].attn.head_dim == 80
    x, y = torch.randn(3, 3, 4, 32, 32), torch.tensor([1, 0, 1])
    P = {k: v.detach().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
    ref = F.cross_entropy(O.forward(x.double(), P, depth=2, num_heads=4, tubelet=2, patch=16, eps=m.blocks[0].norm1.eps), y)
    ref.backward()
    T.set_precision("half")
    try:
        loss = F.cross_entropy(m(x.cuda()), y.cuda())
        loss.backward()
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())  # noqa: E731
        assert abs(loss.item() - ref.item()) < 3e-4
        for k in ("blocks.0.attn.qkv.weight", "blocks.1.attn.proj.weight", "blocks.0.attn.q_bias", "blocks.0.mlp.fc1.weight"):
            assert rel(dict(m.named_parameters())[k].grad, P[k].grad) < 5e-3, k
        md = T.VisionTransformer(attn_drop_rate=0.2, **cfg).cuda().train()
        md.load_state_dict(m.state_dict())
        torch.manual_seed(11); a = md(x.cuda())
        torch.manual_seed(11); b = md(x.cuda())
        torch.manual_seed(12); c = md(x.cuda())
        assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
        md.eval(); m.eval()
        with torch.no_grad():
            assert torch.equal(md(x.cuda()), m(x.cuda()))
    finally:
        T.set_precision("fast")


def test_overflowed_step_is_skipped_on_the_device_and_settled_one_call_later():
    """GradScaler's rules without its host sync: a loss scale that overflows half makes the gradients inf, the fused AdamW kernel skips
    the update by itself (grad_scale 0), and the NEXT call halves the scale and takes the optimizer's step count back before anything
    uses them -- state identical to torch's GradScaler.step / update pair (utils.py:386-412)."""
    import simple_tad_amd as T
    from simple_tad_amd import engine
    import torch.nn.functional as F
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, all_frames=4, tubelet_size=2, num_classes=2,
                            mlp_ratio=4, qkv_bias=True, init_scale=1.0).cuda().train()
    x, y = torch.randn(4, 3, 4, 32, 32).cuda(), torch.tensor([0, 1, 1, 0]).cuda()
    T.set_precision("half")
    try:
        opt = engine.create_optimizer(m, lr=1e-3, weight_decay=0.05)
        sc = engine.NativeScalerWithGradNormCount(m, init_scale=2.0 ** 40, growth_interval=2)
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        n1 = sc(F.cross_entropy(m(x), y), opt, parameters=list(m.parameters()))
        opt.zero_grad()
        torch.cuda.synchronize()
        assert not math.isfinite(float(n1))
        assert all(torch.equal(v, before[k]) for k, v in m.state_dict().items()), "an overflowed step must leave the parameters untouched"
        assert float(opt.exp_avg.abs().max()) == 0.0 and opt.steps == 1 and sc.scale == 2.0 ** 40   # (not settled yet)
        # scale too large again -> second skip; the first one is settled at the start of this call
        n2 = sc(F.cross_entropy(m(x), y), opt, parameters=list(m.parameters()))
        opt.zero_grad()
        assert sc.scale == 2.0 ** 39 and sc.skipped_steps == 1 and opt.steps == 1
        assert sc.state_dict()["scale"] == 2.0 ** 38 and sc.skipped_steps == 2 and opt.steps == 0 and not math.isfinite(float(n2))
        assert all(int(opt.state[p]["step"]) == 0 for p in m.parameters())
        sc.scale = 1024.0  # a workable scale: the step goes through, and two good steps double it (growth_interval 2)
        for i in range(2):
            n = sc(F.cross_entropy(m(x), y), opt, parameters=list(m.parameters()))
            opt.zero_grad()
            assert math.isfinite(float(n))
        assert sc.state_dict()["scale"] == 2048.0 and opt.steps == 2 and sc.skipped_steps == 2
        assert any(not torch.equal(v, before[k]) for k, v in m.state_dict().items())
    finally:
        T.set_precision("fast")


def test_non_finite_norm_under_clipping_without_scaling_is_a_counted_skip():
    """bf16 mode, clip_grad on, a NaN in the gradients: the update is skipped on the device (tad_grad_norm_coef makes the coefficient 0),
    the weights survive, and the skip is accounted for at the next call -- counted, step counts rolled back, scale untouched (ADVICE r05;
    the reference's clip_grad_norm_ would write NaN into every weight here, utils.py:401-404)."""
    import simple_tad_amd as T
    from simple_tad_amd import engine
    import torch.nn.functional as F
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, all_frames=4, tubelet_size=2, num_classes=2,
                            mlp_ratio=4, qkv_bias=True, init_scale=1.0).cuda().train()
    x, y = torch.randn(4, 3, 4, 32, 32).cuda(), torch.tensor([0, 1, 1, 0]).cuda()
    opt = engine.create_optimizer(m, lr=1e-3, weight_decay=0.05)
    sc = engine.NativeScalerWithGradNormCount(m)
    assert not sc.scaling()
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    loss = F.cross_entropy(m(x), y) * float("nan")
    n1 = sc(loss, opt, clip_grad=1.0, parameters=list(m.parameters()))
    opt.zero_grad()
    torch.cuda.synchronize()
    assert not math.isfinite(float(n1))
    assert all(torch.equal(v, before[k]) for k, v in m.state_dict().items()), "a NaN norm must leave the parameters untouched"
    assert opt.steps == 1 and sc.skipped_steps == 0  # (not settled yet)
    n2 = sc(F.cross_entropy(m(x), y), opt, clip_grad=1.0, parameters=list(m.parameters()))
    opt.zero_grad()
    assert math.isfinite(float(n2)) and sc.skipped_steps == 1 and sc.scale == 65536.0
    assert sc.state_dict()["scale"] == 1.0 and opt.steps == 1, "the skipped step's count was taken back, the good step counted"
    assert all(int(opt.state[p]["step"]) == 1 for p in m.parameters())
    assert any(not torch.equal(v, before[k]) for k, v in m.state_dict().items())
