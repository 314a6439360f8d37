"""Seeded random-shape sweep of the HIP kernels against fp64 references on the same bf16-rounded inputs: shapes are drawn so that they land
on and around every plan boundary of the launchers (tile sizes 64 / 128 / 256, the 2048-row switch to the big tiles, the persistent / split-tail
plans above 1.5 tiles per CU, ragged last tiles, widths that are only multiples of 4 / 8, reductions of 64 .. 4096).  The hand-picked shapes
of test_kernels_gpu.py pin the model's own sizes; this file is there to find what nobody picked."""
import numpy as np
import pytest
import torch

from attn_util import prescaled_pair
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-3
BF16_ULP = 2.0 ** -8


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from simple_tad_amd import kernels, _lib
    _lib.load()
    return kernels


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return max(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), ((a - b).norm() / b.norm().clamp_min(1e-30)).item())


def rnd(g, shape, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).float()


def shapes(seed, n, m_choices, n_mult, k_choices):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        M = int(rs.choice(m_choices)) + int(rs.randint(-3, 4))
        N = int(rs.randint(1, 260)) * n_mult
        Kd = int(rs.choice(k_choices))
        out.append((max(M, 1), N, Kd))
    return out


# rows around the tile heights, the small / large plan switch (2048) and the persistent threshold (1.5 tiles per CU: 98 304 rows at N <= 256
# is too much for a sweep, so N up to 1036 with M up to 33 000 gets there: 129 x 5 tiles > 384)
M_CHOICES = [1, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 700, 2047, 2048, 2049, 2600, 5000, 9000, 33000]
K_CHOICES = [64, 128, 192, 320, 768, 1536, 4096]


@pytest.mark.parametrize("M,N,Kd", shapes(101, 14, M_CHOICES, 4, K_CHOICES))
def test_linear_forward_random_shapes(K, M, N, Kd):
    g = torch.Generator().manual_seed(M * 7919 + N * 31 + Kd)
    x, w, b = rnd(g, (M, Kd)), rnd(g, (N, Kd), 0.05), torch.randn(N, generator=g) * 0.1
    xd, wd, bd = x.cuda().to(torch.bfloat16), w.cuda().to(torch.bfloat16), b.cuda()
    ref = x.double() @ w.double().t() + b.double()
    y, _ = K.linear_fwd(xd, wd, bd, out_dtype=torch.float32)
    assert rel(y, ref) <= TOL, ("bias f32", rel(y, ref))
    y16, _ = K.linear_fwd(xd, wd, bd)
    assert rel(y16.float(), ref) <= BF16_ULP, ("bias bf16", rel(y16.float(), ref))
    yg, pre = K.linear_fwd(xd, wd, bd, epilogue=1, want_preact=True)
    assert rel(yg.float(), O.gelu_erf(ref)) <= BF16_ULP and rel(pre.float(), ref) <= BF16_ULP, "gelu / preact"
    res = torch.randn(M, N, generator=g)
    gam = torch.randn(N, generator=g) * 0.3 + 1.0
    rows_per = int(torch.randint(1, max(2, min(M, 2000)), (1,), generator=g))
    rs = (torch.rand((M + rows_per - 1) // rows_per, generator=g) > 0.3).float() * 1.25
    yr, _ = K.linear_fwd(xd, wd, bd, out_dtype=torch.float32, epilogue=2, residual=res.cuda(), gamma=gam.cuda(), rowscale=rs.cuda(), rows_per_scale=rows_per)
    ref_r = res.double() + rs.double().repeat_interleave(rows_per)[:M, None] * gam.double() * ref
    assert rel(yr, ref_r) <= TOL, ("residual", rows_per, rel(yr, ref_r))
    dropped = rs.repeat_interleave(rows_per)[:M] == 0
    assert torch.equal(yr.cpu()[dropped], res[dropped]), "dropped rows must keep the residual bit for bit"


@pytest.mark.parametrize("M,N,Kd", shapes(202, 10, M_CHOICES, 8, K_CHOICES))
def test_linear_backward_random_shapes(K, M, N, Kd):
    """y [M,N] = x [M,Kd] W^T: dx = dy W (also through GELU'), dW = dy^T x, db = column sums"""
    g = torch.Generator().manual_seed(M * 104729 + N * 17 + Kd)
    # (the input-gradient GEMM reduces over N: the wrapper pads an N off the 64-deep K-tile with zero columns)
    dy, w, x = rnd(g, (M, N)), rnd(g, (N, Kd), 0.05), rnd(g, (M, Kd))
    dyd, xd = dy.cuda().to(torch.bfloat16), x.cuda().to(torch.bfloat16)
    wT = K.transpose_cast_bf16(w.cuda())                   # [N, Kd] f32 -> bf16 [Kd, N]
    ref_dx = dy.double() @ w.double()
    dx = K.linear_bwd_input(dyd, wT, out_dtype=torch.float32)
    assert rel(dx, ref_dx) <= TOL, ("dx", rel(dx, ref_dx))
    h = rnd(g, (M, Kd), 1.5)
    hd = h.double().requires_grad_()
    O.gelu_erf(hd).backward(ref_dx)
    dxg = K.linear_bwd_input(dyd, wT, out_dtype=torch.float32, gelu_preact=h.cuda().to(torch.bfloat16))
    assert rel(dxg, hd.grad) <= TOL, ("dx through gelu", rel(dxg, hd.grad))
    dW, db = K.linear_bwd_weight(dyd, xd)
    assert rel(dW, dy.double().t() @ x.double()) <= TOL and rel(db, dy.double().sum(0)) <= TOL, "dW / db"


@pytest.mark.parametrize("seed", range(8))
def test_layernorm_random_shapes(K, seed):
    rs = np.random.RandomState(300 + seed)
    rows = int(rs.choice([1, 3, 4, 5, 31, 33, 200, 1568, 4099]))
    D = int(rs.choice([4, 64, 128, 252, 256, 384, 768, 1000, 1024, 1280, 2048]))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    w, b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    dy = rnd(g, (rows, D))
    xd = x.double().requires_grad_()
    wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
    ref = O.layer_norm(xd, wd, bd, 1e-6)
    ref.backward(dy.double())
    y, mean, rstd = K.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), 1e-6, out_dtype=torch.float32)
    assert rel(y, ref) <= TOL, ("ln fwd", rows, D, rel(y, ref))
    dx, _, dg, db, _ = K.layernorm_bwd(dy.cuda().to(torch.bfloat16), x.cuda(), w.cuda(), mean, rstd)
    assert rel(dx, xd.grad) <= TOL and rel(dg, wd.grad) <= TOL and rel(db, bd.grad) <= TOL, ("ln bwd", rows, D)


@pytest.mark.parametrize("seed", range(8))
def test_attention_random_shapes(K, seed):
    rs = np.random.RandomState(400 + seed)
    B, H = int(rs.randint(1, 4)), int(rs.randint(1, 5))
    N = int(rs.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 193, 392, 500, 1000]))
    d = 80 if seed % 3 == 2 else 64  # every third seed: the "huge" head dim (side images, fifth k-step)
    scale = d ** -0.5
    g = torch.Generator().manual_seed(seed)
    qkv = rnd(g, (B * N, 3 * H * d))
    dout = rnd(g, (B * N, H * d))
    pre = bool(seed & 1)  # odd seeds: the production contract (q third pre-scaled by scale * log2e), even: plain q
    opnd = qkv
    if pre:
        opnd, qkv = prescaled_pair(qkv, B, N, H, scale, lambda t: t.to(torch.bfloat16).float(), d=d)
    q = qkv.double().reshape(B, N, -1).requires_grad_()
    ref = O.attention_core(q, H, scale)
    ref.backward(dout.double().reshape(B, N, -1))
    qd = opnd.cuda().to(torch.bfloat16)
    out32, _ = K.attn_fwd(qd, B, N, H, scale, out_dtype=torch.float32, q_prescaled=pre, d=d)
    assert rel(out32.reshape(B, N, -1), ref) <= 4e-3, ("attn fwd f32", B, N, H, rel(out32.reshape(B, N, -1), ref))  # (P is rounded to bf16 inside: ATT_TOL)
    out, lse = K.attn_fwd(qd, B, N, H, scale, q_prescaled=pre, d=d)
    assert rel(out.float().reshape(B, N, -1), ref) <= 4e-3 + BF16_ULP / 2, ("attn fwd bf16", B, N, H, rel(out.float().reshape(B, N, -1), ref))
    dqkv = K.attn_bwd(qd, out, dout.cuda().to(torch.bfloat16), lse, B, N, H, scale, q_prescaled=pre, d=d)
    e = ((dqkv.float().cpu().double().reshape(B, N, -1) - q.grad).abs().max() / q.grad.abs().max()).item()
    assert e <= 2 * BF16_ULP, ("attn bwd", B, N, H, e)
