"""torch.autograd glue over the HIP kernels.  Each Function mirrors one reference operator
(modeling_finetune.py) and calls only `simple_tad_amd.kernels` (the C ABI) for arithmetic on
activations; torch is used for memory, streams, autograd bookkeeping and a few O(B) / O(D)
scalar-sized tensors (bias concatenation, drop-path masks).

Data flow of one fused Block (training), M = B*N rows:
    x0 f32 --LN1--> xn1 bf16 --qkv GEMM(+bias)--> qkv bf16 [M,3D] --flash attn--> ao bf16 [M,D]
       --proj GEMM (+bias, +residual x0)--> x1 f32 --LN2--> xn2 bf16
       --fc1 GEMM (+bias, GELU; pre-activation kept)--> a bf16 [M,4D] --fc2 GEMM (+bias, +residual x1)--> x2 f32
The residual stream stays f32 (as under torch autocast); GEMM / attention operands are bf16 with f32 accumulation.
"""
from __future__ import annotations

import threading
import weakref
from typing import Optional

import torch

from . import kernels as K
from ._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, TadError

# --------------------------------------------------------------------------- bf16 weight copies
# GEMM operands are bf16 copies of the f32 master weights (torch owns the parameters; cached copies must never outlive
# an optimizer step).  ``p._version`` cannot be the only guard: torch's fused / capturable optimizers update parameters through
# ``_fused_adamw_`` WITHOUT bumping the version counter (measured on torch 2.10: version 0 -> 0 across a fused AdamW step).  So:
#   * a forward that will be differentiated (training) always re-derives its copies from the master weights and drops any cached
#     entry for them (``fresh=True``); backward derives the transposed copy at backward time (weights do not change between the
#     two);
#   * only forwards without autograd (inference) reuse cached copies, keyed by (version, data_ptr), and ``invalidate_weight_cache``
#     (called by this package's optimizer wrapper after every step) clears them explicitly.
_wcache = {}  # id(param) -> {"ref": weakref, "key": (version, data_ptr), kind: tensor}
_MAKERS = {}


def invalidate_weight_cache():
    """Drop every cached bf16 weight copy AND every optimizer-written operand mirror (call after modifying parameters by any means
    that may not bump ``_version``: ``p.data.copy_``, a torch fused optimizer step, EMA swaps).  optim.FusedAdamW calls this and then
    re-registers the mirrors it has just rewritten."""
    global _weight_epoch
    _weight_epoch += 1
    _wcache.clear()
    _mirrors.clear()
    _mirrors_t.clear()


_weight_epoch = 0


def weight_epoch() -> int:
    """bumped by every invalidate_weight_cache(): holders of captured HIP graphs compare it to know when to re-capture"""
    return _weight_epoch


def cached_weight_tensors():
    """every bf16 weight copy currently cached (a captured graph keeps this list so that the addresses it baked in stay allocated)"""
    return [v for ent in _wcache.values() for k, v in ent.items() if k not in ("ref", "key")]


def _w2d(p):
    w2 = p.detach().reshape(p.shape[0], -1)
    if w2.dtype != torch.float32:
        w2 = w2.float()
    return w2 if w2.is_contiguous() else w2.contiguous()


_MAKERS["n"] = lambda p: K.cast_bf16(_w2d(p))                                  # [N,K] bf16
_MAKERS["t"] = lambda p: K.transpose_cast_bf16(_w2d(p))                        # [K,N] bf16
_MAKERS["s"] = lambda p: K.split_bf16x3(_w2d(p), role_b=True)                  # [N,3K] split operand ([hi|lo|hi]), precise Linear
_MAKERS["st"] = lambda p: K.split_bf16x3(_w2d(p).t().contiguous(), role_b=True)  # [K,3N] split operand of W^T, precise dX


# Operand mirrors: optim.FusedAdamW writes bf16(p) for every weight in the same pass that updates p and registers the views
# here; a mirror is served only while the parameter still is the tensor (same storage, same version) the optimizer wrote.
_mirrors = {}    # id(param) -> (weakref, bf16 [N,K] view, version, data_ptr)
_mirrors_t = {}  # id(param) -> (weakref, bf16 [K,N] view of W^T, version, data_ptr)


def register_mirror(p: torch.Tensor, view: torch.Tensor, transposed: bool = False):
    pid = id(p)
    tab = _mirrors_t if transposed else _mirrors
    tab[pid] = (weakref.ref(p, lambda _r, pid=pid, tab=tab: tab.pop(pid, None)), view, p._version, p.data_ptr())


def _mirror_of(p, transposed: bool = False):
    ent = (_mirrors_t if transposed else _mirrors).get(id(p))
    if ent is not None and ent[0]() is p and ent[2] == p._version and ent[3] == p.data_ptr() and ent[1].dtype == K.operand_dtype():
        return ent[1]  # (a mirror written in the other 16-bit format -- the precision mode changed after the optimizer was built -- is not served)
    return None


# Gradient sinks: when a parameter's gradient lives in a flat buffer (flat.FlatSpace, set up by parallel.DataParallel or
# optim.FusedAdamW), the weight-gradient GEMM accumulates straight into ``p.grad`` (the kernel's ``accumulate`` mode) instead
# of materialising dW and letting autograd run ``p.grad += dW`` -- one read-modify-write pass over every weight gradient and
# ~80 small launches per step less.  The Function then returns None for that parameter, so autograd's post-accumulate hooks do
# not fire; the sink's ``notify`` callback (DataParallel's bucket logic) is called instead.
_sinks = {}  # id(param) -> (weakref, flat gradient view, notify | None)


def register_grad_sink(p: torch.Tensor, view: torch.Tensor, notify=None):
    pid = id(p)
    _sinks[pid] = (weakref.ref(p, lambda _r, pid=pid: _sinks.pop(pid, None)), view, notify)


def _sink(p):
    """the flat gradient view of p if p.grad currently IS that view (f32, contiguous), else None"""
    if p is None:
        return None
    ent = _sinks.get(id(p))
    if ent is None or ent[0]() is not p or p.grad is None or p.grad.data_ptr() != ent[1].data_ptr() or torch.is_grad_enabled():
        return None  # (grad mode is on inside backward only under create_graph=True: leave that to autograd)
    return ent


def linear_dw(dy, x, w, b=None, loose_bias=False):
    """(dW, db) of a Linear for autograd.  With gradient sinks on the parameters the gradients are accumulated in place and None is
    returned in their stead.  ``loose_bias``: the column sums of dy are wanted as a tensor although there is no bias Parameter to
    sink them into (qkv: they are split into q_bias / v_bias afterwards)."""
    sw = _sink(w)
    sb = _sink(b) if b is not None else None
    if sw is not None and (b is None or sb is not None):
        db = sb[1] if b is not None else (torch.zeros(w.shape[0], dtype=torch.float32, device=dy.device) if loose_bias else None)
        K.linear_bwd_weight(dy, x, want_bias=db is not None, dW=sw[1].view(w.shape[0], -1), db=db, accumulate=True)
        for ent, prm in ((sw, w), (sb, b)):
            if ent is not None and ent[2] is not None:
                ent[2](prm)
        return None, (db if b is None else None)
    return K.linear_bwd_weight(dy, x, want_bias=(b is not None or loose_bias))


def layernorm_bwd_sunk(dy, x, gamma, mean, rstd, w, b, colsum_param=None, **kw):
    """K.layernorm_bwd whose reductions (dgamma, dbeta and, with ``colsum_param``, the column sums = that bias's gradient) go into
    the parameters' gradient sinks when all of them have one.  Returns (dx, dx_bf16, dgamma|None, dbeta|None, colsum|None)."""
    ents = [_sink(w), _sink(b)] + ([_sink(colsum_param)] if colsum_param is not None else [])
    if all(e is not None for e in ents):
        dx, dxb, _, _, _ = K.layernorm_bwd(dy, x, gamma, mean, rstd, want_colsum=colsum_param is not None,
                                           into=(ents[0][1], ents[1][1], ents[2][1] if colsum_param is not None else None), **kw)
        for ent, prm in zip(ents, (w, b, colsum_param)):
            if ent[2] is not None:
                ent[2](prm)
        return dx, dxb, None, None, None
    return K.layernorm_bwd(dy, x, gamma, mean, rstd, want_colsum=colsum_param is not None, **kw)


def _cached(p: torch.Tensor, kind: str, fresh: bool = False):
    pid = id(p)
    if kind in ("n", "t"):
        m = _mirror_of(p, kind == "t")
        if m is not None:
            return m
    if fresh:
        _wcache.pop(pid, None)
        return _MAKERS[kind](p)
    ent = _wcache.get(pid)
    key = (p._version, p.data_ptr())
    if ent is None or ent["ref"]() is not p or ent["key"] != key:
        ent = {"ref": weakref.ref(p, lambda _r, pid=pid: _wcache.pop(pid, None)), "key": key}
        _wcache[pid] = ent
    if kind not in ent:
        ent[kind] = _MAKERS[kind](p)
    return ent[kind]


def w_bf16(p, fresh: bool = False):
    return _cached(p, "n", fresh)


def wT_bf16(p, fresh: bool = False):
    return _cached(p, "t", fresh)


def _f32c(t: Optional[torch.Tensor]):
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _need_gpu(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise TadError(f"{what}: input is on {x.device}; the MI355X path runs HIP kernels only (no CPU fallback). "
                       "Move the model and inputs to a GPU.")


def _qkv_bias(q_bias, v_bias):
    if q_bias is None:
        return None
    return torch.cat((q_bias.detach(), torch.zeros_like(v_bias), v_bias.detach())).float().contiguous()


# --------------------------------------------------------------------------- precise mode (parity gate)
def _cached_split(p: torch.Tensor, fresh: bool = False):
    return _cached(p, "s", fresh)


# --------------------------------------------------------------------------- precision mode
_PRECISION = "fast"


def set_precision(mode: str):
    """Numerics of the MFMA path.
    "fast":    bfloat16 operands, f32 accumulation / residual stream / statistics -- the benchmarked default.
    "half":    IEEE-half operands through the same kernels (tad_*_f16): 8x smaller operand rounding at the same MFMA rate.  This is the
               reference's own arithmetic -- torch.cuda.amp.autocast() is float16 on CUDA (engine_for_finetuning.py:67) -- and like
               there the backward pass needs loss scaling: engine.NativeScalerWithGradNormCount scales the loss, and the fused AdamW
               removes the scale and skips overflowed steps (utils.py:386-412).
    "precise": split-bf16 (hi + lo, three MFMA products) Linears, f32 attention and f32 activations -- the parity gate."""
    global _PRECISION
    if mode not in ("fast", "half", "precise"):
        raise ValueError(mode)
    if mode != _PRECISION:
        invalidate_weight_cache()  # cached operand copies are in the old format
    _PRECISION = mode
    K.set_operand_dtype(torch.float16 if mode == "half" else torch.bfloat16)


def get_precision() -> str:
    return _PRECISION


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise TadError('precision "precise" is forward-only (parity gate); wrap the call in torch.no_grad()')


def _cached_split_T(p: torch.Tensor, fresh: bool = False):
    return _cached(p, "st", fresh)


def precise_dx(dy, weight):
    """dx [M,K] f32 = dy [M,N] @ W [N,K] with split operands (reduction over 3N); backward only, so the copy is always re-derived"""
    dx, _ = K.linear_fwd(K.split_bf16x3(dy, role_b=False), _cached_split_T(weight, True), None, out_dtype=torch.float32)
    return dx


def precise_dw(dy, x, want_bias=True):
    """dW [N,K] = dy^T x with both operands split and stacked along the reduction (row) dimension; db = column sums of dy"""
    dW, _ = K.linear_bwd_weight(K.split_bf16x3(dy, role_b=False, stack=True), K.split_bf16x3(x, role_b=True, stack=True), want_bias=False)
    return dW, (K.colsum_f32(dy) if want_bias else None)


def precise_linear(x2d, weight, bias, epilogue=EPI_BIAS, residual=None, rowscale=None, rows_per_scale=1, fresh=False):
    xs = K.split_bf16x3(x2d, role_b=False)
    y, _ = K.linear_fwd(xs, _cached_split(weight, fresh), _f32c(bias), out_dtype=torch.float32, epilogue=epilogue, residual=residual,
                        rowscale=rowscale, rows_per_scale=rows_per_scale)
    return y


def precise_attention(x2d, B, N, qkv_w, q_bias, v_bias, proj_w, proj_b, H, scale, residual=None):
    qkv = precise_linear(x2d, qkv_w, _qkv_bias(q_bias, v_bias))
    ao, _ = K.attn_fwd_f32(qkv, B, N, H, scale, d=head_dim_of(qkv_w, H))
    return precise_linear(ao, proj_w, proj_b, EPI_BIAS_RESIDUAL if residual is not None else EPI_BIAS, residual=residual)


def precise_mlp(x2d, fc1_w, fc1_b, fc2_w, fc2_b, residual=None):
    a = precise_linear(x2d, fc1_w, fc1_b, EPI_BIAS_GELU)
    return precise_linear(a, fc2_w, fc2_b, EPI_BIAS_RESIDUAL if residual is not None else EPI_BIAS, residual=residual)


def precise_block(x, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, H, scale, eps):
    return PreciseBlockFn.apply(x, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, H, scale, eps)


def precise_patch_embed(x, weight, bias, pos, tubelet, patch):
    return PrecisePatchEmbedFn.apply(x, weight, bias, pos, tubelet, patch)




# --------------------------------------------------------------------------- grad mode seen by the CALLER of a Function
# Inside Function.forward autograd has switched grad mode off and ctx.needs_input_grad is True for every Parameter even under
# torch.no_grad(), so "will this forward be differentiated?" (decides whether cached bf16 weight copies may be reused, whether
# pre-activations / statistics are kept) has to be sampled before apply().
_outer = threading.local()


class _Fn(torch.autograd.Function):
    @classmethod
    def apply(cls, *args, **kwargs):
        prev = getattr(_outer, "grad", None)
        _outer.grad = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _outer.grad = prev


def _differentiated(ctx) -> bool:
    g = getattr(_outer, "grad", None)
    return (True if g is None else g) and any(ctx.needs_input_grad)

# --------------------------------------------------------------------------- LayerNorm
class LayerNormFn(_Fn):
    """nn.LayerNorm(D, eps) on f32 rows -> f32 (modeling_finetune.py:143,149,270)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _need_gpu(x, "LayerNorm")
        xc = _f32c(x)
        w, b = _f32c(weight), _f32c(bias)
        y, mean, rstd = K.layernorm_fwd(xc.reshape(-1, xc.shape[-1]), w, b, eps, out_dtype=torch.float32)
        ctx.save_for_backward(xc, w, mean, rstd)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        xc, w, mean, rstd = ctx.saved_tensors
        D = xc.shape[-1]
        dx, _, dg, db, _ = K.layernorm_bwd(_f32c(dy).reshape(-1, D), xc.reshape(-1, D), w, mean, rstd)
        return dx.reshape(xc.shape), dg, db, None


# --------------------------------------------------------------------------- PatchEmbed (+ pos_embed)
# The implicit patch embedding (tad_patch_embed_fwd_implicit) is taken by forwards without a backward pass once the problem fills the chip: measured
# (tools/bench_kernels.py --only patch, us, explicit / implicit): 32 clips D = 768 264 / 245, D = 384 193 / 152, ONE clip (78 tiles of 128 x 128 on 256
# CUs, 24 dependent K-tiles each) 23 / 33 -- below two tiles per CU the im2col + GEMM pair is faster
IMPLICIT_PATCH_EMBED_MIN_TILES = 512


class PatchEmbedFn(_Fn):
    """Conv3d(k=s=(tub,p,p)) + flatten/transpose (+ sinusoid pos_embed) (modeling_finetune.py:181-190, 312-313)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pos, tubelet, patch):
        _need_gpu(x, "PatchEmbed")
        xc = _f32c(x)
        # patch sizes whose K = C*tub*p*p is not a multiple of 64 (ViT-L/14: 1176): the patch matrix and the weight operand carry zero
        # columns up to tad_patch_embed_ldk (1216); a pure configuration change for the callers
        ldk = K.patch_embed_ldk(xc.shape[1], tubelet, patch)
        ntok_ = (xc.shape[2] // tubelet) * (xc.shape[3] // patch) * (xc.shape[4] // patch)
        if not _differentiated(ctx) and -(-xc.shape[0] * ntok_ // 128) * -(-weight.shape[0] // 128) >= IMPLICIT_PATCH_EMBED_MIN_TILES:
            # nothing is kept for a backward pass (eval / no_grad / inference): the implicit GEMM reads the clip itself and
            # writes no patch matrix; bit-identical to the explicit form below, which the training step needs for its weight gradient
            out = K.patch_embed_fwd_implicit(xc, w_bf16(weight, False), _f32c(bias), _f32c(pos), tubelet, patch)
            if out is not None:
                ctx.params = (weight, bias)
                return out
        out, cols = K.patch_embed_fwd(xc, K.pad_k(w_bf16(weight, _differentiated(ctx)), ldk), _f32c(bias), _f32c(pos), tubelet, patch)
        ctx.save_for_backward(cols)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        weight, bias = ctx.params
        D = dy.shape[-1]
        dyb = K.cast_bf16(_f32c(dy).reshape(-1, D))
        kw = weight.numel() // weight.shape[0]
        if cols.shape[1] != kw:  # padded K: the gradient of the padding columns is dropped (no in-place sink for a [D, ldk] result)
            dWp, db = K.linear_bwd_weight(dyb, cols, want_bias=bias is not None)
            return None, dWp[:, :kw].reshape(weight.shape), db, None, None, None
        dW, db = linear_dw(dyb, cols, weight, bias)
        return None, (None if dW is None else dW.reshape(weight.shape)), db, None, None, None


class PatchEmbedU8Fn(_Fn):
    """PatchEmbed on uint8 frames [B,T,H,W,3]: normalisation (run_inference.py:15-34; dota.py:443-460) fused into the im2col."""

    @staticmethod
    def forward(ctx, frames, weight, bias, pos, tubelet, patch, mean, std, bgr, t_offset):
        _need_gpu(frames, "PatchEmbed")
        fr = frames if frames.is_contiguous() else frames.contiguous()
        B, T, H, W, _ = fr.shape
        cols = K.im2col_tubelets_u8(fr, tubelet, patch, mean, std, bgr, t_offset)   # row stride tad_patch_embed_ldk (zero-padded for /14)
        ntok = (T // tubelet) * (H // patch) * (W // patch)
        out = K.patch_embed_gemm(cols, K.pad_k(w_bf16(weight, _differentiated(ctx)), cols.shape[1]), _f32c(bias), _f32c(pos), ntok)
        ctx.save_for_backward(cols)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        weight, bias = ctx.params
        dyb = K.cast_bf16(_f32c(dy).reshape(-1, dy.shape[-1]))
        kw = weight.numel() // weight.shape[0]
        if cols.shape[1] != kw:  # padded K (see PatchEmbedFn.backward)
            dWp, db = K.linear_bwd_weight(dyb, cols, want_bias=bias is not None)
            return None, dWp[:, :kw].reshape(weight.shape), db, None, None, None, None, None, None, None
        dW, db = linear_dw(dyb, cols, weight, bias)
        return None, (None if dW is None else dW.reshape(weight.shape)), db, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- attention / mlp cores (2-D tensors)
def head_dim_of(qkv_w, H):
    return qkv_w.shape[0] // (3 * H)


# The attention forward also writes what the 16-bit rounding of its output dropped (77 MB per ViT-B layer at 32 clips), and the
# backward takes delta = rowsum(dO * O) of the unrounded output: see tad_attn_bwd.  Off: delta from the rounded output, as flash-attn.
_attn_exact_delta = True


_attn_q_prescale = True  # the qkv Linear writes q * scale * log2(e) (False: plain q, the kernels scale their fragments; for A/B runs)


def set_attn_q_prescale(on: bool):
    """takes effect at the next forward; every autograd node remembers the contract its own forward ran under (ctx.qpre), so a toggle
    between a forward and its backward cannot mis-read the saved qkv"""
    global _attn_q_prescale
    _attn_q_prescale = bool(on)


def get_attn_q_prescale() -> bool:
    return _attn_q_prescale


def set_attn_exact_delta(on: bool):
    global _attn_exact_delta
    _attn_exact_delta = bool(on)


_attn_hd80_f32 = False


def set_attn_hd80_f32(on: bool):
    """A/B switch: head_dim 80 attention through the exact-f32 MFMA kernels (the round-3 route) instead of the 16-bit ones"""
    global _attn_hd80_f32
    _attn_hd80_f32 = bool(on)


def _attn_fwd_core(xn, qkv_w, q_bias, v_bias, B, N, H, scale, train, drop=(0.0, 0)):
    """returns qkv, attention output, (lse, rounding residual of the output | None).  drop = (p, seed) of attention dropout."""
    qb = None if q_bias is None else _f32c(q_bias.detach())
    vb = None if v_bias is None else _f32c(v_bias.detach())
    hd = head_dim_of(qkv_w, H)
    if hd not in (64, 80):
        raise TadError(f"attention: head_dim {hd} has no kernel (64 and 80 do)")
    if hd == 80 and _attn_hd80_f32:
        # The round-3 route for the "huge" configurations (head_dim 80, modeling_finetune.py:390-398), kept for A/B runs
        # (set_attn_hd80_f32): the Linears stay on the 16-bit MFMA GEMMs, the scaled-dot-product core runs through the exact-f32 MFMA
        # kernels (csrc/attn_f32.hip) on an f32 qkv.  Since round 4 head_dim 80 has 16-bit kernels of its own (the branch below).
        qkv = K.linear_fwd_qkv(xn, w_bf16(qkv_w, train), qb, vb, out_dtype=torch.float32)
        ao32, lse = K.attn_fwd_f32(qkv, B, N, H, scale, want_lse=train, d=hd, drop_p=drop[0], seed=drop[1])
        return qkv, K.cast_bf16(ao32), (lse, None)
    # `q = q * self.scale` (modeling_finetune.py:96) rides in the qkv Linear's epilogue, together with log2(e): the attention kernels
    # get their scores from the matrix pipe in log2 units and spend no vector instruction on the scale (the q third of `qkv` is NOT
    # the plain q; only tad_attn_fwd / tad_attn_bwd read it)
    qkv = K.linear_fwd_qkv(xn, w_bf16(qkv_w, train), qb, vb, out_dtype=None, q_prescale=K.q_prescale_of(scale) if _attn_q_prescale else 1.0)
    r = K.attn_fwd(qkv, B, N, H, scale, out_dtype=None, want_lse=train, want_lo=train and _attn_exact_delta, q_prescaled=_attn_q_prescale,
                   drop_p=drop[0], seed=drop[1], d=hd)
    return qkv, r[0], (r[1], r[2] if len(r) > 2 else None)


def _attn_bwd_core(d_ao, xn, qkv, ao, lse, qkv_w, has_qkv_bias, B, N, H, scale, dx_dtype, qv_params=None, ao_lo=None, drop=(0.0, 0),
                   q_prescaled=None, pair=None):
    """returns dxn, dWqkv, dq_bias, dv_bias (None for what went into gradient sinks).  q_prescaled: the q contract of the forward
    that produced `qkv` (ctx.qpre).  pair = (dy, x, w) of ANOTHER bias-less weight gradient over the same rows with the same K that is due
    (the Block's proj: dy = the 16-bit gradient of its output, x = the attention output): a fifth value is returned then, that Linear's dW
    (None if it went into its gradient sink).  With sinks on all parameters the two weight gradients run as ONE launch
    (K.linear_bwd_weight_pair: a 768 x 768 gradient alone fills the chip with 9 tiles x 28 reduction shares, the pair with 36 x 7)."""
    if q_prescaled is None:
        q_prescaled = _attn_q_prescale
    if qkv.dtype == torch.float32:  # generic head dim (see _attn_fwd_core)
        dqkv = K.cast_bf16(K.attn_bwd_f32(qkv, ao.float(), d_ao.float(), lse, B, N, H, scale, d=head_dim_of(qkv_w, H), drop_p=drop[0],
                                          seed=drop[1]))
    else:
        dqkv = K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, scale, out_lo=ao_lo, q_prescaled=q_prescaled, drop_p=drop[0], seed=drop[1],
                          d=head_dim_of(qkv_w, H))
    dxn = K.linear_bwd_input(dqkv, wT_bf16(qkv_w, True), out_dtype=dx_dtype)

    def done(*r):  # (+ the pair's weight gradient by itself when it did not ride along)
        return r if pair is None else r + (linear_dw(pair[0], pair[1], pair[2])[0],)

    if has_qkv_bias and qv_params is not None:
        ents = [_sink(qkv_w), _sink(qv_params[0]), _sink(qv_params[1])]
        if all(e is not None for e in ents):
            # dW, dq_bias and dv_bias accumulate straight into the flat gradient buffer: no [3D] temporary, no slicing copies, no
            # autograd accumulation kernels
            prms = (qkv_w,) + tuple(qv_params)
            sp = _sink(pair[2]) if pair is not None else None
            paired = sp is not None and pair[0].dtype == dqkv.dtype and pair[1].shape == xn.shape and pair[0].shape[0] == dqkv.shape[0]
            if paired:
                K.linear_bwd_weight_pair(dqkv, xn, ents[0][1].view(qkv_w.shape[0], -1), ents[1][1], ents[2][1], pair[0], pair[1],
                                         sp[1].view(pair[2].shape[0], -1), accumulate=True)
                ents, prms = ents + [sp], prms + (pair[2],)
            else:
                K.linear_bwd_weight_qkv(dqkv, xn, ents[0][1].view(qkv_w.shape[0], -1), ents[1][1], ents[2][1], accumulate=True)
            for ent, prm in zip(ents, prms):
                if ent[2] is not None:
                    ent[2](prm)
            if paired:
                return dxn, None, None, None, None
            return done(dxn, None, None, None)
    dWqkv, dbqkv = linear_dw(dqkv, xn, qkv_w, None, loose_bias=has_qkv_bias)
    if has_qkv_bias:
        AH = dbqkv.numel() // 3
        return done(dxn, dWqkv, dbqkv[:AH].clone(), dbqkv[2 * AH:].clone())
    return done(dxn, dWqkv, None, None)


class AttentionFn(_Fn):
    """Attention.forward (modeling_finetune.py:86-134): qkv Linear, scaled-dot-product space-time attention, proj."""

    @staticmethod
    def forward(ctx, x, qkv_w, q_bias, v_bias, proj_w, proj_b, H, scale, drop_p=0.0, drop_seed=0):
        _need_gpu(x, "Attention")
        B, N, C = x.shape
        train = _differentiated(ctx)
        ctx.drop = (float(drop_p), int(drop_seed))
        xb = K.cast_bf16(_f32c(x).reshape(B * N, C))
        qkv, ao, (lse, ao_lo) = _attn_fwd_core(xb, qkv_w, q_bias, v_bias, B, N, H, scale, train, ctx.drop)
        ctx.qpre = _attn_q_prescale  # (the contract `qkv` was written under: read back by the backward)
        y, _ = K.linear_fwd(ao, w_bf16(proj_w, train), _f32c(proj_b), out_dtype=torch.float32)
        if train:
            ctx.save_for_backward(xb, qkv, ao, lse, qkv_w, proj_w, ao_lo)
        ctx.meta = (B, N, H, scale, q_bias is not None, proj_b is not None)
        ctx.proj_b = proj_b
        ctx.qv = (q_bias, v_bias)
        return y.reshape(B, N, -1)

    @staticmethod
    def backward(ctx, dy):
        xb, qkv, ao, lse, qkv_w, proj_w, ao_lo = ctx.saved_tensors
        B, N, H, scale, has_qb, has_pb = ctx.meta
        dyb = K.cast_bf16(_f32c(dy).reshape(B * N, -1))
        d_ao = K.linear_bwd_input(dyb, wT_bf16(proj_w, True))
        dWp, dbp = linear_dw(dyb, ao, proj_w, ctx.proj_b)
        dx, dWqkv, dqb, dvb = _attn_bwd_core(d_ao, xb, qkv, ao, lse, qkv_w, has_qb, B, N, H, scale, torch.float32, ctx.qv, ao_lo=ao_lo,
                                             drop=ctx.drop, q_prescaled=ctx.qpre)
        return dx.reshape(B, N, -1), dWqkv, dqb, dvb, dWp, dbp, None, None, None, None


class MlpFn(_Fn):
    """Mlp.forward (modeling_finetune.py:47-54): fc2(GELU_erf(fc1(x)))."""

    @staticmethod
    def forward(ctx, x, fc1_w, fc1_b, fc2_w, fc2_b):
        _need_gpu(x, "Mlp")
        shp = x.shape
        train = _differentiated(ctx)
        xb = K.cast_bf16(_f32c(x).reshape(-1, shp[-1]))
        a, h = K.linear_fwd(xb, w_bf16(fc1_w, train), _f32c(fc1_b), out_dtype=None, epilogue=EPI_BIAS_GELU, want_preact=train)
        y, _ = K.linear_fwd(a, w_bf16(fc2_w, train), _f32c(fc2_b), out_dtype=torch.float32)
        if train:
            ctx.save_for_backward(xb, h, a, fc1_w, fc2_w)
        ctx.meta = (shp, fc1_b is not None, fc2_b is not None)
        ctx.biases = (fc1_b, fc2_b)
        return y.reshape(*shp[:-1], -1)

    @staticmethod
    def backward(ctx, dy):
        xb, h, a, fc1_w, fc2_w = ctx.saved_tensors
        shp, has_b1, has_b2 = ctx.meta
        dyb = K.cast_bf16(_f32c(dy).reshape(-1, dy.shape[-1]))
        dh = K.linear_bwd_input(dyb, wT_bf16(fc2_w, True), gelu_preact=h)
        dW2, db2 = linear_dw(dyb, a, fc2_w, ctx.biases[1])
        dx = K.linear_bwd_input(dh, wT_bf16(fc1_w, True), out_dtype=torch.float32)
        dW1, db1 = linear_dw(dh, xb, fc1_w, ctx.biases[0])
        return dx.reshape(shp), dW1, db1, dW2, db2


# --------------------------------------------------------------------------- fused Block
class _ChainLink:
    """Hand-off between two consecutive fused Blocks, one object per Block forward.  Block i's backward starts by casting the incoming
    residual-stream gradient to bf16 scaled by ITS drop-path scale dp2 -- a pass over an f32 [M, D] tensor that block i+1's LayerNorm
    backward, which produces that very gradient, can emit on the side.

    Forward: ``block_apply`` creates the link of block i, stores it in block i's ctx and hangs it on block i's OUTPUT TENSOR OBJECT;
    block i+1 finds it on its input (``_tad_link``) and keeps it as ``ctx.prev_link``.  Anything in between that produces a new tensor
    (checkpointing, a hook, ``x + 0``) drops the attribute, so the next block simply has no predecessor.
    Backward: block i+1 deposits (bf16 copy, the f32 gradient tensor it returns, that tensor's version) in ITS prev_link -- the
    object only block i's ctx of the SAME forward holds; block i takes the copy only if the gradient autograd hands it is still that
    storage, shape and version.  A second consumer of block i's output makes autograd either allocate a new sum (pointer differs)
    or accumulate in place (version differs): both miss and fall back to the cast pass.  Nothing is keyed by raw pointers in a
    process-wide table, so a stale entry of an earlier iteration can never be picked up by a later one."""
    __slots__ = ("dp2", "deposit", "__weakref__")

    def __init__(self, dp2):
        self.dp2 = dp2
        self.deposit = None


class _ChainState:
    def __init__(self):
        self.enabled = True
        self.hits = 0      # hand-offs taken (tests)
        self.pending = 0   # deposits made and not yet consumed or dropped (tests: must be 0 after a backward pass)


_chain = _ChainState()


def reset_block_chain():
    """kept for API compatibility: the hand-off state lives in per-forward link objects and needs no reset"""


def set_block_chain(enabled: bool):
    """switch the Block-to-Block bf16 gradient hand-off off / on (results are bit-identical either way; for tests and timing)"""
    _chain.enabled = bool(enabled)


def block_apply(x, *args):
    """BlockFn.apply plus the chain bookkeeping on the input / output tensor objects"""
    prev = getattr(x, "_tad_link", None) if _chain.enabled else None
    dp2 = args[14]
    link = _ChainLink(None if dp2 is None else _f32c(dp2)) if (_chain.enabled and torch.is_grad_enabled()) else None
    out = BlockFn.apply(x, *args, prev, link)
    if link is not None and out.requires_grad:
        out._tad_link = link
    return out


class BlockFn(_Fn):
    """Block.forward without layer-scale (modeling_finetune.py:159-163):
         x = x + dp1 * attn(norm1(x));  x = x + dp2 * mlp(norm2(x))
    dp1/dp2 are optional per-sample drop-path scales [B] (mask / keep_prob) or None."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, dp1, dp2, H, scale,
                eps, prev_link=None, link=None):
        _need_gpu(x, "Block")
        B, N, D = x.shape
        M = B * N
        train = _differentiated(ctx)
        x0 = _f32c(x).reshape(M, D)
        g1, b1, g2, b2 = _f32c(n1w), _f32c(n1b), _f32c(n2w), _f32c(n2b)
        xn1, mean1, rstd1 = K.layernorm_fwd(x0, g1, b1, eps, save_stats=train)
        qkv, ao, (lse, ao_lo) = _attn_fwd_core(xn1, qkv_w, q_bias, v_bias, B, N, H, scale, train)
        ctx.qpre = _attn_q_prescale  # (the contract `qkv` was written under: read back by the backward)
        x1, _ = K.linear_fwd(ao, w_bf16(proj_w, train), _f32c(proj_b), out_dtype=torch.float32, epilogue=EPI_BIAS_RESIDUAL, residual=x0,
                             rowscale=_f32c(dp1), rows_per_scale=N)
        xn2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, eps, save_stats=train)
        a, h = K.linear_fwd(xn2, w_bf16(fc1_w, train), _f32c(fc1_b), out_dtype=None, epilogue=EPI_BIAS_GELU, want_preact=train)
        x2, _ = K.linear_fwd(a, w_bf16(fc2_w, train), _f32c(fc2_b), out_dtype=torch.float32, epilogue=EPI_BIAS_RESIDUAL, residual=x1,
                             rowscale=_f32c(dp2), rows_per_scale=N)
        if train:
            ctx.save_for_backward(x0, g1, mean1, rstd1, xn1, qkv, ao, lse, x1, g2, mean2, rstd2, xn2, h, a, qkv_w, proj_w, fc1_w, fc2_w,
                                  dp1 if dp1 is None else _f32c(dp1), dp2 if dp2 is None else _f32c(dp2), ao_lo)
        ctx.meta = (B, N, D, H, scale, q_bias is not None)
        ctx.biases = (proj_b, fc1_b, fc2_b)
        ctx.norms = (n1w, n1b, n2w, n2b)
        ctx.qv = (q_bias, v_bias)
        # block chain (_ChainLink): where my backward deposits the bf16 gradient for the block before me / takes the one made for me
        ctx.prev_link = prev_link if train else None
        ctx.link = link if train else None
        return x2.reshape(B, N, D)

    @staticmethod
    def backward(ctx, g):
        (x0, g1, mean1, rstd1, xn1, qkv, ao, lse, x1, g2, mean2, rstd2, xn2, h, a, qkv_w, proj_w, fc1_w, fc2_w, dp1,
         dp2, ao_lo) = ctx.saved_tensors
        proj_b, fc1_b, fc2_b = ctx.biases
        B, N, D, H, scale, has_qb = ctx.meta
        M = B * N
        g = _f32c(g).reshape(M, D)
        # ---- MLP branch
        gb = None
        dep = ctx.link.deposit if ctx.link is not None else None
        if dep is not None:
            ctx.link.deposit = None
            _chain.pending -= 1
            cand, ref, ver = dep
            if ref.data_ptr() == g.data_ptr() and ref._version == ver and g._version == ver and cand.shape == g.shape:
                gb = cand
                _chain.hits += 1
        if gb is None:
            gb = K.cast_bf16(g) if dp2 is None else K.scale_cast_bf16(g, None, dp2, N)
        dh = K.linear_bwd_input(gb, wT_bf16(fc2_w, True), gelu_preact=h)
        dW2, db2 = linear_dw(gb, a, fc2_w, fc2_b)
        dxn2 = K.linear_bwd_input(dh, wT_bf16(fc1_w, True))
        dW1, db1 = linear_dw(dh, xn2, fc1_w, fc1_b)
        # LN2 backward also emits what the attention branch needs: bf16(drop-path scale * residual-stream grad) and its column
        # sums (= proj bias gradient), so neither a cast pass nor a bias reduction in the dW GEMM is needed
        n1w, n1b, n2w, n2b = ctx.norms
        gmid, gpb, dg2, dbeta2, dbp = layernorm_bwd_sunk(dxn2, x1, g2, mean2, rstd2, n2w, n2b, colsum_param=proj_b, dres=g, want_bf16=True,
                                                        rowscale=dp1, rows_per_scale=N)
        # ---- attention branch
        d_ao = K.linear_bwd_input(gpb, wT_bf16(proj_w, True))
        # (the proj weight gradient rides with the qkv one: one launch for both, _attn_bwd_core)
        dxn1, dWqkv, dqb, dvb, dWp = _attn_bwd_core(d_ao, xn1, qkv, ao, lse, qkv_w, has_qb, B, N, H, scale, None, ctx.qv, ao_lo=ao_lo, q_prescaled=ctx.qpre,
                                                     pair=(gpb, ao, proj_w))
        prev = ctx.prev_link
        if prev is not None:
            gin, ginb, dg1, dbeta1, _ = layernorm_bwd_sunk(dxn1, x0, g1, mean1, rstd1, n1w, n1b, dres=gmid, want_bf16=True, rowscale=prev.dp2,
                                                           rows_per_scale=N)
            if prev.deposit is None:
                _chain.pending += 1
            prev.deposit = (ginb, gin, gin._version)
        else:
            gin, _, dg1, dbeta1, _ = layernorm_bwd_sunk(dxn1, x0, g1, mean1, rstd1, n1w, n1b, dres=gmid)
        return (gin.reshape(B, N, D), dg1, dbeta1, dWqkv, dqb, dvb, dWp, dbp, dg2, dbeta2, dW1, db1, dW2, db2, None, None, None, None,
                None, None, None)


# --------------------------------------------------------------------------- plain Linear (f32 rows in / out)
class LinearFn(_Fn):
    """nn.Linear on f32 rows through the bf16 MFMA GEMM (encoder_to_decoder and the decoder head, modeling_pretrain.py:163,269)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, "Linear")
        shp = x.shape
        train = _differentiated(ctx)
        xb = K.cast_bf16(_f32c(x).reshape(-1, shp[-1]))
        y, _ = K.linear_fwd(xb, w_bf16(weight, train), _f32c(bias), out_dtype=torch.float32)
        if train:
            ctx.save_for_backward(xb)
        ctx.params = (weight, bias)
        ctx.shp = shp
        return y.reshape(*shp[:-1], -1)

    @staticmethod
    def backward(ctx, dy):
        (xb,) = ctx.saved_tensors
        weight, bias = ctx.params
        dyb = K.cast_bf16(_f32c(dy).reshape(-1, dy.shape[-1]))
        dx = K.linear_bwd_input(dyb, wT_bf16(weight, True), out_dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dW, db = linear_dw(dyb, xb, weight, bias)
        return (None if dx is None else dx.reshape(ctx.shp)), dW, db


# --------------------------------------------------------------------------- MAE pre-training path
class GatherRowsFn(_Fn):
    """x[~mask].reshape(B, -1, C) (modeling_pretrain.py:98) with precomputed row indices (b*N + visible token)."""

    @staticmethod
    def forward(ctx, x, idx, B):
        _need_gpu(x, "token gather")
        Bx, N, D = x.shape
        ctx.save_for_backward(idx)
        ctx.shape = (Bx, N, D)
        return K.gather_rows(_f32c(x).reshape(Bx * N, D), idx).reshape(B, -1, D)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        Bx, N, D = ctx.shape
        return K.scatter_rows(_f32c(g).reshape(-1, D), idx, Bx * N).reshape(Bx, N, D), None, None


class MaeAssembleFn(_Fn):
    """cat([x_vis + pos[vis], mask_token + pos[masked]], dim=1) (modeling_pretrain.py:283-287); the positional table is a constant"""

    @staticmethod
    def forward(ctx, x_vis, mask_token, pos, vis_idx, mask_idx):
        _need_gpu(x_vis, "MAE decoder input")
        B, Nv, D = x_vis.shape
        ctx.meta = (B, Nv, mask_idx.numel() // B, D)
        return K.mae_assemble(_f32c(x_vis).reshape(B * Nv, D), _f32c(mask_token).reshape(-1), _f32c(pos).reshape(-1, D), vis_idx, mask_idx, B)

    @staticmethod
    def backward(ctx, g):
        B, Nv, Nm, D = ctx.meta
        g = _f32c(g)
        d_xv = g[:, :Nv].contiguous()
        d_tok = K.colsum_window_f32(g, Nv, Nm).reshape(1, 1, D)  # (the f64 verification kernel on a contiguous copy took 2.3 ms here)
        return d_xv, d_tok, None, None, None


class MseLossFn(_Fn):
    """nn.MSELoss()(outputs, labels) (engine_for_pretraining.py:27,70): loss and d(loss)/d(outputs) in one pass"""

    @staticmethod
    def forward(ctx, pred, target):
        _need_gpu(pred, "MSE loss")
        loss, grad = K.mse_loss(_f32c(pred), _f32c(target), want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).reshape(ctx.shape), None


# --------------------------------------------------------------------------- mean-pool
class MeanPoolFn(_Fn):
    """x.mean(1) over tokens (modeling_finetune.py:325-326)."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "mean-pool")
        ctx.n = x.shape[1]
        return K.meanpool_fwd(_f32c(x))

    @staticmethod
    def backward(ctx, dy):
        dx, _ = K.meanpool_bwd(_f32c(dy), ctx.n)
        return dx


# --------------------------------------------------------------------------- precise mode: autograd
class PreciseBlockFn(_Fn):
    """Block forward/backward in the precise mode (f32 activations, split-bf16 Linears, f32 attention): the gradient side of the
    parity gate.  Same data flow as BlockFn; no drop-path (verification runs use drop_path_rate = 0)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, H, scale, eps):
        _need_gpu(x, "Block")
        B, N, D = x.shape
        M = B * N
        fr = _differentiated(ctx)  # differentiated forward: never reuse cached weight copies (see _cached)
        x0 = _f32c(x).reshape(M, D)
        g1, b1, g2, b2 = _f32c(n1w), _f32c(n1b), _f32c(n2w), _f32c(n2b)
        xn1, mean1, rstd1 = K.layernorm_fwd(x0, g1, b1, eps, out_dtype=torch.float32)
        qkv = precise_linear(xn1, qkv_w, _qkv_bias(q_bias, v_bias), fresh=fr)
        ao, lse = K.attn_fwd_f32(qkv, B, N, H, scale, want_lse=True, d=head_dim_of(qkv_w, H))
        x1 = precise_linear(ao, proj_w, proj_b, EPI_BIAS_RESIDUAL, residual=x0, fresh=fr)
        xn2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, eps, out_dtype=torch.float32)
        h = precise_linear(xn2, fc1_w, fc1_b, fresh=fr)
        a = K.gelu_f32(h)
        x2 = precise_linear(a, fc2_w, fc2_b, EPI_BIAS_RESIDUAL, residual=x1, fresh=fr)
        ctx.save_for_backward(x0, g1, mean1, rstd1, xn1, qkv, ao, lse, x1, g2, mean2, rstd2, xn2, h, a, qkv_w, proj_w, fc1_w, fc2_w)
        ctx.meta = (B, N, D, H, scale, q_bias is not None)
        return x2.reshape(B, N, D)

    @staticmethod
    def backward(ctx, g):
        x0, g1, mean1, rstd1, xn1, qkv, ao, lse, x1, g2, mean2, rstd2, xn2, h, a, qkv_w, proj_w, fc1_w, fc2_w = ctx.saved_tensors
        B, N, D, H, scale, has_qb = ctx.meta
        g = _f32c(g).reshape(B * N, D)
        dh = K.gelu_bwd_f32(precise_dx(g, fc2_w), h)
        dW2, db2 = precise_dw(g, a)
        dxn2 = precise_dx(dh, fc1_w)
        dW1, db1 = precise_dw(dh, xn2)
        gmid, _, dg2, dbeta2, _ = K.layernorm_bwd(dxn2, x1, g2, mean2, rstd2, dres=g)
        d_ao = precise_dx(gmid, proj_w)
        dWp, dbp = precise_dw(gmid, ao)
        dqkv = K.attn_bwd_f32(qkv, ao, d_ao, lse, B, N, H, scale, d=head_dim_of(qkv_w, H))
        dxn1 = precise_dx(dqkv, qkv_w)
        dWqkv, dbqkv = precise_dw(dqkv, xn1, want_bias=has_qb)
        dqb = dvb = None
        if has_qb:
            AH = dbqkv.numel() // 3
            dqb, dvb = dbqkv[:AH].clone(), dbqkv[2 * AH:].clone()
        gin, _, dg1, dbeta1, _ = K.layernorm_bwd(dxn1, x0, g1, mean1, rstd1, dres=gmid)
        return (gin.reshape(B, N, D), dg1, dbeta1, dWqkv, dqb, dvb, dWp, dbp, dg2, dbeta2, dW1, db1, dW2, db2, None, None, None)


class PrecisePatchEmbedFn(_Fn):
    @staticmethod
    def forward(ctx, x, weight, bias, pos, tubelet, patch):
        _need_gpu(x, "PatchEmbed")
        B = x.shape[0]
        cols = K.im2col_tubelets_f32(_f32c(x), tubelet, patch)
        ntok = cols.shape[0] // B
        res = _f32c(pos).repeat(B, 1) if pos is not None else None
        kw = weight.numel() // weight.shape[0]
        ws = (_cached_split(weight, _differentiated(ctx)) if cols.shape[1] == kw
              else K.split_bf16x3(K.pad_k(_w2d(weight), cols.shape[1]).contiguous(), role_b=True))  # zero-padded K (patch 14)
        y, _ = K.linear_fwd(K.split_bf16x3(cols, role_b=False), ws, _f32c(bias),
                            out_dtype=torch.float32, epilogue=EPI_BIAS_RESIDUAL if res is not None else EPI_BIAS, residual=res)
        ctx.save_for_backward(cols)
        ctx.wshape = weight.shape
        ctx.has_bias = bias is not None
        return y.reshape(B, ntok, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        dW, db = precise_dw(_f32c(dy).reshape(-1, dy.shape[-1]), cols, want_bias=ctx.has_bias)
        kw = 1
        for d in ctx.wshape[1:]:
            kw *= d
        return None, dW[:, :kw].reshape(ctx.wshape), db, None, None, None
