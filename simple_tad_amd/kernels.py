"""Tensor-level wrappers over the C ABI: validate shapes on the host, pass raw device
pointers + the current HIP stream.  No arithmetic happens in Python or in torch here.

Every function requires CUDA(=HIP) tensors and raises otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, TAD_BF16, TAD_F16, TAD_F32, check

_workspaces = {}


class LaunchProfiler:
    """Optional per-launch timing with HIP events on the launching stream (used by bench.py for the live roofline figure).
    ``record(kernel, flops, bytes)`` brackets one C-ABI call; ``summary()`` synchronises and aggregates per kernel class."""

    def __init__(self, only=None, stride=1):
        self.items = []
        self.only = set(only) if only else None  # restrict to these kernel classes (keeps the timed region undisturbed)
        # A timed HIP event pair costs ~20 us of queue time on this platform (measured: 192 pairs per step = +2 ms on a 55 ms
        # step), so the bench samples every ``stride``-th launch of a class; a stride coprime with the number of launches per
        # step visits every GEMM shape equally often.
        self.stride = max(1, int(stride))
        self.seen = {}

    def wants(self, kernel):
        if self.only is not None and kernel not in self.only:
            return False
        n = self.seen.get(kernel, 0)
        self.seen[kernel] = n + 1
        return n % self.stride == 0

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, start, kernel, flops, nbytes, kernel_launches=1):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.items.append((kernel, start, e, flops, nbytes, kernel_launches))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kernel, s, e, flops, nbytes, nk in self.items:
            d = out.setdefault(kernel, {"launches": 0, "calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += nk
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


_prof = None


def set_profiler(p):
    global _prof
    _prof = p


class _timed:
    __slots__ = ("k", "f", "b", "s", "n0")

    def __init__(self, kernel, flops=0.0, nbytes=0.0):
        self.k, self.f, self.b = kernel, flops, nbytes

    def __enter__(self):
        self.s = _prof.begin() if (_prof is not None and _prof.wants(self.k)) else None
        if self.s is not None and self.k == "gemm_nt":
            self.n0 = _lib.load().tad_linear_kernel_launches()

    def __exit__(self, *a):
        if self.s is not None:
            # a Linear call is one gemm_nt kernel launch, or two under the split-tail plan: count kernels, not calls
            n = (_lib.load().tad_linear_kernel_launches() - self.n0) if self.k == "gemm_nt" else 1
            _prof.end(self.s, self.k, self.f, self.b, n)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.TadError(f"{name}: expected a GPU tensor (the MI355X path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise _lib.TadError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.TadError(f"{name}: tensor must be contiguous")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TAD_F32
    if t.dtype == torch.bfloat16:
        return TAD_BF16
    if t.dtype == torch.float16:
        return TAD_F16
    raise _lib.TadError(f"unsupported dtype {t.dtype}")


# ----------------------------------------------------------------------------- 16-bit operand format
# Every kernel that touches GEMM / attention operands exists for bfloat16 (tad_*) and for IEEE half (tad_*_f16) -- the same
# kernels compiled for the other format (include/tad_mi355x.h).  A call's format is the dtype of its 16-bit tensors; functions that
# CREATE 16-bit tensors from f32 (casts, LayerNorm, im2col ...) take it from ``dtype=`` or, by default, from the process-wide operand
# format below (ops.set_precision selects it: "fast" -> bfloat16, "half" -> float16).
OP16_DTYPES = (torch.bfloat16, torch.float16)
_op16 = torch.bfloat16


def set_operand_dtype(dtype) -> None:
    global _op16
    if dtype not in OP16_DTYPES:
        raise ValueError(f"operand dtype must be torch.bfloat16 or torch.float16, got {dtype}")
    _op16 = dtype


def operand_dtype():
    return _op16


def _fn(name: str, dtype):
    """the entry point `name` for 16-bit operands of `dtype` (its _f16 twin for torch.float16)"""
    return getattr(_lib.load(), _lib.F16_TWINS[name] if dtype == torch.float16 else name)


def _req16(t: torch.Tensor, name: str, like=None):
    """a contiguous GPU tensor in one of the two 16-bit operand formats (the same one as ``like`` if given); returns its dtype"""
    if not t.is_cuda:
        raise _lib.TadError(f"{name}: expected a GPU tensor (the MI355X path has no CPU fallback), got {t.device}")
    if t.dtype not in OP16_DTYPES or (like is not None and t.dtype != like):
        raise _lib.TadError(f"{name}: expected {like if like is not None else 'bfloat16 or float16'}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.TadError(f"{name}: tensor must be contiguous")
    return t.dtype


def _out16(out_dtype, op):
    """output dtype of a call with 16-bit operands of format `op`: f32, or that same 16-bit format (None = the operands' own)"""
    if out_dtype is None or out_dtype in OP16_DTYPES:
        if out_dtype is not None and out_dtype != op:
            raise _lib.TadError(f"16-bit output {out_dtype} does not match the operands' format {op}")
        return op
    if out_dtype != torch.float32:
        raise _lib.TadError(f"unsupported output dtype {out_dtype}")
    return out_dtype


def _ws_key(device, stream=None):
    dev = device.index if device.index is not None else torch.cuda.current_device()
    return (dev, (torch.cuda.current_stream(device) if stream is None else stream).cuda_stream)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): kernels on one stream run in issue order, so consecutive ops share it; another
    stream (a capture stream) gets its own and never aliases it.  A buffer that is outgrown is retired, not freed, for as long as
    its (device, stream) entry lives -- HIP graphs captured on that stream hold raw pointers into it.  Growth is geometric, so the
    retired buffers together stay below the size of the live one.  Holders of captured graphs warm up ON THE CAPTURE STREAM (so the
    buffer the capture uses exists before the capture starts), keep `workspace_refs` of that stream next to their graphs and call
    `release_workspace` when the stream goes away (inference.SlidingWindow, bench --graph)."""
    key = _ws_key(device)
    ent = _workspaces.get(key)
    if ent is None:
        ent = _workspaces[key] = {"ws": None, "retired": []}
    ws = ent["ws"]
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            ent["retired"].append(ws)
        ws = torch.empty(max(int(nbytes), 2 * (ws.numel() if ws is not None else 0), 1 << 20), dtype=torch.uint8, device=device)
        ent["ws"] = ws
    return ws


def workspace_refs(device, stream=None):
    """every scratch buffer (live and retired) of one (device, stream): what a graph captured on that stream may point into"""
    ent = _workspaces.get(_ws_key(device, stream))
    return [] if ent is None else [t for t in [ent["ws"], ent.get("lin"), *ent["retired"]] if t is not None]


def release_workspace(device, stream=None) -> None:
    """forget the scratch buffers of one (device, stream); they are freed once the last holder of `workspace_refs` lets go"""
    _workspaces.pop(_ws_key(device, stream), None)


# ----------------------------------------------------------------------------- casts
def cast_op16(x: torch.Tensor, out: Optional[torch.Tensor] = None, dtype=None) -> torch.Tensor:
    """f32 -> the 16-bit operand format (``out``'s, ``dtype``, or the process-wide one)"""
    _req(x, torch.float32, "cast_op16.x")
    if out is None:
        out = torch.empty(x.shape, dtype=dtype or _op16, device=x.device)
    op = _req16(out, "cast_op16.out")
    with _timed("cast", 0.0, 6.0 * x.numel()):
        check(_fn("tad_cast_f32_bf16", op)(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "tad_cast_f32_bf16")
    return out


def transpose_cast_op16(w: torch.Tensor, dtype=None) -> torch.Tensor:
    """w [R,C] f32 -> [C,R] in the 16-bit operand format"""
    _req(w, torch.float32, "transpose_cast.w")
    R, Cc = w.shape
    out = torch.empty((Cc, R), dtype=dtype or _op16, device=w.device)
    check(_fn("tad_transpose_cast_f32_bf16", out.dtype)(w.data_ptr(), out.data_ptr(), R, Cc, _stream()), "tad_transpose_cast")
    return out


cast_bf16, transpose_cast_bf16 = cast_op16, transpose_cast_op16  # (names from before the half format existed)


def transpose_table(mats):
    """Host-side tile list for transpose_bf16_batched: mats = [(element offset, R, C)], every matrix stored [R,C] row-major at
    ``offset`` of the source buffer and written [C,R] at the same offset of the destination.  Returns an int32 [n_tiles, 8] tensor."""
    rows = []
    for off, R, C in mats:
        if R % 8 or C % 8 or off % 8:
            raise _lib.TadError(f"transpose_table: matrix [{R},{C}] at {off} is not made of whole 16-byte chunks")
        for r0 in range(0, R, 64):
            for c0 in range(0, C, 64):
                rows.append((off + r0 * C + c0, off + c0 * R + r0, C, R, min(64, R - r0), min(64, C - c0), 0, 0))
    import numpy as np
    arr = np.asarray(rows, dtype=np.int64).reshape(-1, 8)
    if arr.size and int(arr[:, :2].max()) >= (1 << 32):
        raise _lib.TadError("transpose_table: offsets exceed 2^32 elements")
    return torch.from_numpy(arr.astype(np.uint32).view(np.int32))


def transpose_bf16_batched(src, dst, table):
    """16-bit transposes: moves bit patterns, so one entry point serves both operand formats"""
    _req16(dst, "transpose_batched.dst", like=_req16(src, "transpose_batched.src"))
    if table.dtype != torch.int32 or not table.is_cuda or table.dim() != 2 or table.shape[1] != 8 or not table.is_contiguous():
        raise _lib.TadError("transpose_bf16_batched: table must be a contiguous int32 [n_tiles, 8] device tensor")
    with _timed("cast", 0.0, 4.0 * 4096 * table.shape[0]):
        check(_lib.load().tad_transpose_bf16_batched(src.data_ptr(), dst.data_ptr(), table.data_ptr(), table.shape[0], _stream()),
              "tad_transpose_bf16_batched")
    return dst


def scale_cast_op16(x, gamma=None, rowscale=None, rows_per_scale=1, dtype=None):
    _req(x, torch.float32, "scale_cast.x")
    M, N = x.shape
    out = torch.empty((M, N), dtype=dtype or _op16, device=x.device)
    with _timed("cast", 0.0, 6.0 * M * N):
        check(_fn("tad_scale_cast_bf16", out.dtype)(x.data_ptr(), out.data_ptr(), _p(gamma), _p(rowscale), int(rows_per_scale), M, N,
                                                    _stream()), "tad_scale_cast_bf16")
    return out


scale_cast_bf16 = scale_cast_op16


# ----------------------------------------------------------------------------- patch embed
def patch_embed_ldk(C: int, tubelet: int, patch: int) -> int:
    """row stride of the patch matrix / the bf16 weight (K rounded up to 64; == K for /16)"""
    return int(_lib.load().tad_patch_embed_ldk(int(C), int(tubelet), int(patch)))


def pad_k(w: torch.Tensor, ldk: int) -> torch.Tensor:
    """[N, K] -> [N, ldk] with zero columns (the weight operand of a patch-embed GEMM whose K is not a multiple of 64)"""
    return w if w.shape[1] == ldk else torch.nn.functional.pad(w, (0, ldk - w.shape[1]))


def im2col_tubelets(x: torch.Tensor, tubelet: int, patch: int, dtype=None) -> torch.Tensor:
    _req(x, torch.float32, "im2col.x")
    B, Cc, T, H, W = x.shape
    ntok = (T // tubelet) * (H // patch) * (W // patch)
    cols = torch.empty((B * ntok, patch_embed_ldk(Cc, tubelet, patch)), dtype=dtype or _op16, device=x.device)
    check(_fn("tad_im2col_tubelets", cols.dtype)(x.data_ptr(), cols.data_ptr(), B, Cc, T, H, W, tubelet, patch, _stream()),
          "tad_im2col_tubelets")
    return cols


def im2col_tubelets_u8(frames: torch.Tensor, tubelet: int, patch: int, mean, std, bgr: bool = False, t_offset: int = 0, dtype=None) -> torch.Tensor:
    """frames [B,T,H,W,3] uint8 -> normalised bf16 patch matrix [B*N, 3*tub*p*p] (tad_im2col_tubelets_u8)"""
    import ctypes as C
    _req(frames, torch.uint8, "im2col_u8.frames")
    if frames.dim() != 5 or frames.shape[-1] != 3:
        raise _lib.TadError(f"im2col_u8: frames must be [B,T,H,W,3] uint8, got {tuple(frames.shape)}")
    B, T, H, W, _ = frames.shape
    ntok = (T // tubelet) * (H // patch) * (W // patch)
    cols = torch.empty((B * ntok, patch_embed_ldk(3, tubelet, patch)), dtype=dtype or _op16, device=frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    with _timed("im2col_u8", 0.0, float(frames.numel()) + 2.0 * cols.numel()):
        check(_fn("tad_im2col_tubelets_u8", cols.dtype)(frames.data_ptr(), cols.data_ptr(), B, T, H, W, tubelet, patch, m, s, int(bool(bgr)),
                                                 int(t_offset), _stream()), "tad_im2col_tubelets_u8")
    return cols


def patch_embed_gemm(cols, w_bf16, bias, pos, ntok: int):
    """cols [B*ntok, K] bf16 -> out [B, ntok, D] f32 = cols w^T + bias (+ pos [ntok, D] broadcast over the batch)"""
    op = _req16(cols, "patch_embed_gemm.cols")
    _req16(w_bf16, "patch_embed_gemm.w", like=op)
    M, Kd = cols.shape
    D, K2 = w_bf16.shape
    if Kd != K2 or M % ntok:
        raise _lib.TadError(f"patch_embed_gemm: cols {tuple(cols.shape)} vs weight {tuple(w_bf16.shape)}, ntok {ntok}")
    if pos is not None:
        _req(pos, torch.float32, "patch_embed_gemm.pos")
        if tuple(pos.shape) != (ntok, D):
            raise _lib.TadError(f"patch_embed_gemm: pos_embed shape {tuple(pos.shape)} != {(ntok, D)}")
    if bias is not None:
        _req(bias, torch.float32, "patch_embed_gemm.bias")
    out = torch.empty((M // ntok, ntok, D), dtype=torch.float32, device=cols.device)
    with _timed("patch_embed_fwd", 2.0 * M * D * Kd, 2.0 * M * Kd + 4.0 * M * D):
        check(_fn("tad_patch_embed_gemm", op)(cols.data_ptr(), w_bf16.data_ptr(), _p(bias), _p(pos), out.data_ptr(), M, ntok, D, Kd,
                                               _stream()), "tad_patch_embed_gemm")
    return out


def patch_embed_fwd(x, w_bf16, bias, pos, tubelet: int, patch: int):
    """x [B,C,T,H,W] f32, w_bf16 [D,K], bias [D] f32|None, pos [N,D] f32|None -> (out [B,N,D] f32, cols [B*N,K] bf16)"""
    _req(x, torch.float32, "patch_embed.x")
    op = _req16(w_bf16, "patch_embed.w")
    B, Cc, T, H, W = x.shape
    D, K = w_bf16.shape
    if K != patch_embed_ldk(Cc, tubelet, patch):
        raise _lib.TadError(f"patch_embed: weight row stride {K} does not match tad_patch_embed_ldk = {patch_embed_ldk(Cc, tubelet, patch)} "
                            f"(C*tub*p*p = {Cc * tubelet * patch * patch} rounded up to 64; kernels.pad_k)")
    ntok = (T // tubelet) * (H // patch) * (W // patch)
    if pos is not None:
        _req(pos, torch.float32, "patch_embed.pos")
        if tuple(pos.shape) != (ntok, D):
            raise _lib.TadError(f"patch_embed: pos_embed shape {tuple(pos.shape)} != {(ntok, D)}")
    if bias is not None:
        _req(bias, torch.float32, "patch_embed.bias")
    cols = torch.empty((B * ntok, K), dtype=op, device=x.device)
    out = torch.empty((B, ntok, D), dtype=torch.float32, device=x.device)
    with _timed("patch_embed_fwd", 2.0 * B * ntok * D * K, 4.0 * x.numel() + 2.0 * 2 * B * ntok * K + 4.0 * B * ntok * D):
        check(_fn("tad_patch_embed_fwd", op)(x.data_ptr(), w_bf16.data_ptr(), _p(bias), _p(pos), out.data_ptr(), cols.data_ptr(),
                                              B, Cc, T, H, W, tubelet, patch, D, _stream()), "tad_patch_embed_fwd")
    return out, cols


def patch_embed_fwd_implicit(x, w_bf16, bias, pos, tubelet: int, patch: int):
    """the same forward without a patch matrix (tad_patch_embed_fwd_implicit: the x operand read straight from the f32 clip); for forwards that keep
    nothing for a backward pass.  Returns out [B,N,D] f32, or None when the implicit kernel does not take the case (patch != 16, a clip of 2 GiB
    and more): the caller then uses patch_embed_fwd"""
    _req(x, torch.float32, "patch_embed.x")
    op = _req16(w_bf16, "patch_embed.w")
    B, Cc, T, H, W = x.shape
    D, K = w_bf16.shape
    if patch != 16 or K != Cc * tubelet * 256 or x.numel() * 4 >= (1 << 31) or D % 4 or T % tubelet or H % 16 or W % 16:
        return None
    ntok = (T // tubelet) * (H // patch) * (W // patch)
    if pos is not None:
        _req(pos, torch.float32, "patch_embed.pos")
        if tuple(pos.shape) != (ntok, D):
            raise _lib.TadError(f"patch_embed: pos_embed shape {tuple(pos.shape)} != {(ntok, D)}")
    if bias is not None:
        _req(bias, torch.float32, "patch_embed.bias")
    out = torch.empty((B, ntok, D), dtype=torch.float32, device=x.device)
    with _timed("patch_embed_fwd", 2.0 * B * ntok * D * K, 4.0 * x.numel() + 4.0 * B * ntok * D):
        check(_fn("tad_patch_embed_fwd_implicit", op)(x.data_ptr(), w_bf16.data_ptr(), _p(bias), _p(pos), out.data_ptr(), B, Cc, T, H, W, tubelet, patch, D,
                                                       _stream()), "tad_patch_embed_fwd_implicit")
    return out


# ----------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps: float, out_dtype=None, save_stats=True):
    """out_dtype: torch.float32, or a 16-bit operand format (None = the process-wide one)"""
    _req(x, torch.float32, "layernorm.x")
    _req(gamma, torch.float32, "layernorm.gamma")
    _req(beta, torch.float32, "layernorm.beta")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=out_dtype or _op16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    with _timed("layernorm_fwd", 0.0, rows * D * (4.0 + y.element_size())):
        check(_fn("tad_layernorm_fwd", y.dtype if y.dtype in OP16_DTYPES else _op16)(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _dt(y), _p(mean), _p(rstd),
                                            rows, D, float(eps), _stream()), "tad_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, want_bf16=False, want_colsum=False, rowscale=None, rows_per_scale=1,
                  into=None):
    """returns dx f32, dx_bf16|None, dgamma, dbeta, colsum_dx|None.  rowscale [rows/rows_per_scale]: per-sample scale applied to the
    bf16 copy and the column sums (drop-path).  into=(dgamma, dbeta, colsum|None): accumulate the three reductions into these
    existing f32 tensors (gradient sinks) instead of returning fresh ones."""
    _req(x, torch.float32, "layernorm_bwd.x")
    if dy.dtype not in (torch.float32,) + OP16_DTYPES or not dy.is_contiguous():
        raise _lib.TadError("layernorm_bwd.dy: must be contiguous f32, bf16 or f16")
    op = dy.dtype if dy.dtype in OP16_DTYPES else _op16  # format of the 16-bit copy of dx (and of dy, if it is 16-bit)
    D = x.shape[-1]
    rows = x.numel() // D
    if dres is not None:
        _req(dres, torch.float32, "layernorm_bwd.dres")
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dxb = torch.empty(x.shape, dtype=op, device=x.device) if want_bf16 else None
    if into is not None:
        dg, db, cs = into
        for t in (dg, db) + ((cs,) if want_colsum else ()):
            _req(t, torch.float32, "layernorm_bwd.into")
            assert t.numel() == D
        if not want_colsum:
            cs = None
    else:
        dg = torch.empty(D, dtype=torch.float32, device=x.device)
        db = torch.empty(D, dtype=torch.float32, device=x.device)
        cs = torch.empty(D, dtype=torch.float32, device=x.device) if want_colsum else None
    if rowscale is not None:
        _req(rowscale, torch.float32, "layernorm_bwd.rowscale")
        assert rowscale.numel() * rows_per_scale >= rows
    lib = _lib.load()
    nbytes = lib.tad_layernorm_bwd_workspace_bytes(rows, D)
    ws = workspace(nbytes, x.device)
    with _timed("layernorm_bwd", 0.0, rows * D * (dy.element_size() + 4.0 + 4.0 + (4.0 if dres is not None else 0.0) + (2.0 if want_bf16 else 0.0))):
        check(_fn("tad_layernorm_bwd", op)(dy.data_ptr(), _dt(dy), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dres),
                                    dx.data_ptr(), _p(dxb), dg.data_ptr(), db.data_ptr(), _p(cs), _p(rowscale), int(rows_per_scale),
                                    int(into is not None), ws.data_ptr(), ws.numel(), rows, D, _stream()), "tad_layernorm_bwd")
    return dx, dxb, dg, db, cs


# ----------------------------------------------------------------------------- linear
def linear_fwd(x, w, bias=None, out_dtype=None, epilogue=EPI_BIAS, want_preact=False, residual=None, gamma=None,
               rowscale=None, rows_per_scale=1):
    """x [M,K], w [N,K] (both bf16 or both f16) -> y [M,N] (out_dtype: f32, or None = the operands' format); returns (y, preact|None)"""
    op = _req16(x, "linear.x")
    _req16(w, "linear.w", like=op)
    out_dtype = _out16(out_dtype, op)
    if x.shape[1] == w.shape[1] and x.shape[1] % 64:
        x, w = _pad_reduction(x, w)  # reduction length off the 64-deep K-tile (a 1176-wide MAE decoder head of a /14 model): zero columns
    M, K = x.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.TadError(f"linear: x K={K} vs w K={K2}")
    if bias is not None:
        _req(bias, torch.float32, "linear.bias")
        assert bias.numel() == N
    if residual is not None:
        _req(residual, torch.float32, "linear.residual")
        assert tuple(residual.shape) == (M, N)
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    pre = torch.empty((M, N), dtype=op, device=x.device) if want_preact else None
    ws, wsb = _linear_ws(M, N, K, x.device)
    # algorithmic bytes of the launch: both operands once, the output once, plus what the epilogue reads (f32 residual) / writes (16-bit pre-activation)
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K) + y.element_size() * M * N + (4.0 * M * N if residual is not None else 0.0)
                + (2.0 * M * N if want_preact else 0.0)):
        check(_fn("tad_linear_fwd", op)(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), _dt(y), epilogue, _p(pre), _p(residual),
                                         _p(gamma), _p(rowscale), int(rows_per_scale), ws, wsb, M, N, K, _stream()), "tad_linear_fwd")
    return y, pre


# Split-K tails of the Linear GEMMs (include/tad_mi355x.h: tad_linear_fwd's `ws`): a per-(device, stream) scratch buffer is handed to the
# Linear calls unless switched off here (experiments).  parallel.DataParallel leaves it ON at world sizes > 1: it pins the deferred
# three-launch form (tad_linear_tuning("splitk_defer", 1)), which is ordered by kernel boundaries and needs no co-residency beside an
# overlapped RCCL kernel; only the in-launch combine ("splitk_defer", 0) does.
_linear_splitk = True


def set_linear_splitk(on: bool):
    global _linear_splitk
    _linear_splitk = bool(on)


def _linear_ws(M, N, K, device):
    if not _linear_splitk:
        return None, 0
    nbytes = _lib.load().tad_linear_workspace_bytes(M, N, K)
    if not nbytes:
        return None, 0
    return linear_workspace(nbytes, device).data_ptr(), nbytes


def _pad_reduction(a, b, mult: int = 64):
    """zero-pad the shared last (reduction) dimension of two operands to a multiple of the GEMM's K-tile: the product is unchanged"""
    pad = (-a.shape[1]) % mult
    return torch.nn.functional.pad(a, (0, pad)).contiguous(), torch.nn.functional.pad(b, (0, pad)).contiguous()


LINEAR_TUNING_DEFAULTS = dict(persistent=1, direct_epilogue=1, split_tail=1, splitk_tail=1, group_m=0, variant=0, tn_pdeep=0, splitk_defer=1, tn_w4=1, w4_plain=640, w4_epilogues=4, tail_192=1, short_k=1, tn_pair=1)


def linear_workspace(nbytes: int, device) -> torch.Tensor:
    """scratch of the split-K Linear tails, one per (device, stream): a buffer of its own (neither buffer's growth retires the other)
    inside the SAME per-(device, stream) entry as `workspace`, so that `workspace_refs` reports it to holders of captured graphs and
    `release_workspace` drops it with the rest"""
    key = _ws_key(device)
    ent = _workspaces.get(key)
    if ent is None:
        ent = _workspaces[key] = {"ws": None, "retired": []}
    buf = ent.get("lin")
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            ent["retired"].append(buf)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ent["lin"] = buf
    return buf


def linear_kernel_launches() -> int:
    """gemm_nt kernel launches issued so far (tad_linear_kernel_launches)"""
    return int(_lib.load().tad_linear_kernel_launches())


def linear_tuning(**knobs):
    """Scheduling knobs of the Linear GEMMs (include/tad_mi355x.h: tad_linear_tuning); timing only, never results."""
    for k, v in knobs.items():
        check(_lib.load().tad_linear_tuning(k.encode(), int(v)), f"tad_linear_tuning({k}={v})")


def linear_tuning_get(key: str) -> int:
    """current value of a tad_linear_tuning knob"""
    v = C.c_int(0)
    check(_lib.load().tad_linear_tuning_get(key.encode(), C.byref(v)), f"tad_linear_tuning_get({key})")
    return int(v.value)


def linear_bwd_input(dy, wT, out_dtype=None, gelu_preact=None):
    """dy [M,N], wT [K,N] (both bf16 or both f16) -> dx [M,K]"""
    op = _req16(dy, "linear_bwd_input.dy")
    _req16(wT, "linear_bwd_input.wT", like=op)
    out_dtype = _out16(out_dtype, op)
    if dy.shape[1] == wT.shape[1] and dy.shape[1] % 64:
        dy, wT = _pad_reduction(dy, wT)
    M, N = dy.shape
    K, N2 = wT.shape
    assert N == N2, (N, N2)
    if gelu_preact is not None:
        _req16(gelu_preact, "linear_bwd_input.gelu_preact", like=op)
        assert tuple(gelu_preact.shape) == (M, K)
    dx = torch.empty((M, K), dtype=out_dtype, device=dy.device)
    ws, wsb = _linear_ws(M, K, N, dy.device)
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * N + N * K) + dx.element_size() * M * K + (2.0 * M * K if gelu_preact is not None else 0.0)):
        check(_fn("tad_linear_bwd_input", op)(dy.data_ptr(), wT.data_ptr(), dx.data_ptr(), _dt(dx), _p(gelu_preact), ws, wsb, M, N, K,
                                               _stream()), "tad_linear_bwd_input")
    return dx


def linear_bwd_weight(dy, x, want_bias=True, dW=None, db=None, accumulate=False):
    """dy [M,N] bf16, x [M,K] bf16 -> dW [N,K] f32, db [N] f32|None"""
    op = _req16(dy, "linear_bwd_weight.dy")
    _req16(x, "linear_bwd_weight.x", like=op)
    M, N = dy.shape
    M2, K = x.shape
    assert M == M2
    if dW is None:
        dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        accumulate = False
    if want_bias and db is None:
        db = torch.empty(N, dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    ws = workspace(lib.tad_linear_bwd_weight_workspace_bytes(M, N, K), dy.device)
    with _timed("gemm_tn", 2.0 * M * N * K, 2.0 * (M * N + M * K) + 4.0 * N * K):
        check(_fn("tad_linear_bwd_weight", op)(dy.data_ptr(), x.data_ptr(), dW.data_ptr(), _p(db) if want_bias else None, int(accumulate),
                                        ws.data_ptr(), ws.numel(), M, N, K, _stream()), "tad_linear_bwd_weight")
    return dW, (db if want_bias else None)


LOG2E = 1.4426950408889634


def q_prescale_of(scale: float) -> float:
    """the factor the attention kernels expect on a pre-scaled q: softmax scale * log2(e) (scores in log2 units)"""
    return float(scale) * LOG2E


def linear_fwd_qkv(x, w, q_bias, v_bias, out_dtype=None, q_prescale: float = 1.0):
    """qkv Linear with bias = cat(q_bias, 0, v_bias) (modeling_finetune.py:89-92) taken from the two parameters directly.
    q_prescale != 1: the q third of the output is multiplied by it before the rounding to the output type (`q = q * self.scale`,
    modeling_finetune.py:96, folded into the Linear; attn_fwd / attn_bwd then take q_prescaled=True)"""
    op = _req16(x, "linear_qkv.x")
    _req16(w, "linear_qkv.w", like=op)
    out_dtype = _out16(out_dtype, op)
    M, K = x.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.TadError(f"linear_qkv: x K={K} vs w K={K2}")
    if q_bias is not None:
        _req(q_bias, torch.float32, "linear_qkv.q_bias")
        _req(v_bias, torch.float32, "linear_qkv.v_bias")
        assert q_bias.numel() == v_bias.numel() == N // 3
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K) + y.element_size() * M * N):
        check(_fn("tad_linear_fwd_qkv", op)(x.data_ptr(), w.data_ptr(), _p(q_bias), _p(v_bias), y.data_ptr(), _dt(y), float(q_prescale), M, N, K,
                                            _stream()), "tad_linear_fwd_qkv")
    return y


def linear_bwd_weight_qkv(dy, x, dW, dq_bias, dv_bias, accumulate):
    """weight gradient of the qkv Linear with the bias column sums split into dq_bias / dv_bias [N/3] (in place)"""
    op = _req16(dy, "linear_bwd_weight_qkv.dy")
    _req16(x, "linear_bwd_weight_qkv.x", like=op)
    M, N = dy.shape
    M2, K = x.shape
    assert M == M2 and dq_bias.numel() == dv_bias.numel() == N // 3
    lib = _lib.load()
    ws = workspace(lib.tad_linear_bwd_weight_workspace_bytes(M, N, K), dy.device)
    with _timed("gemm_tn", 2.0 * M * N * K, 2.0 * (M * N + M * K) + 4.0 * N * K):
        check(_fn("tad_linear_bwd_weight_qkv", op)(dy.data_ptr(), x.data_ptr(), dW.data_ptr(), dq_bias.data_ptr(), dv_bias.data_ptr(), int(accumulate),
                                            ws.data_ptr(), ws.numel(), M, N, K, _stream()), "tad_linear_bwd_weight_qkv")
    return dW


def linear_bwd_weight_pair(dy1, x1, dW1, db1, db1b, dy2, x2, dW2, accumulate):
    """two weight gradients over the same rows with the same K in one call (tad_linear_bwd_weight_pair): dW1 [N1,K] (+)= dy1^T x1 with its
    bias column sums (db1 [N1] | None; db1b given: the first / last third to db1 / db1b), dW2 [N2,K] (+)= dy2^T x2.  In place."""
    op = _req16(dy1, "linear_bwd_weight_pair.dy1")
    for t, n in ((x1, "x1"), (dy2, "dy2"), (x2, "x2")):
        _req16(t, "linear_bwd_weight_pair." + n, like=op)
    M, N1 = dy1.shape
    N2 = dy2.shape[1]
    K = x1.shape[1]
    assert x1.shape[0] == M and dy2.shape[0] == M and x2.shape == (M, K), "linear_bwd_weight_pair: the two problems share M and K"
    assert dW1.shape == (N1, K) and dW2.shape == (N2, K) and dW1.dtype == dW2.dtype == torch.float32
    lib = _lib.load()
    ws = workspace(max(lib.tad_linear_bwd_weight_workspace_bytes(M, n, K) for n in (N1, N2, N1 + N2)), dy1.device)
    with _timed("gemm_tn", 2.0 * M * (N1 + N2) * K, 2.0 * (M * (N1 + N2) + 2 * M * K) + 4.0 * (N1 + N2) * K):
        check(_fn("tad_linear_bwd_weight_pair", op)(dy1.data_ptr(), x1.data_ptr(), dW1.data_ptr(), _p(db1), _p(db1b), N1, dy2.data_ptr(), x2.data_ptr(),
                                                    dW2.data_ptr(), N2, int(accumulate), ws.data_ptr(), ws.numel(), M, K, _stream()),
              "tad_linear_bwd_weight_pair")


def colsum_bf16(a, out=None):
    op = _req16(a, "colsum.a")
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    ws = workspace(lib.tad_colsum_workspace_bytes(M, N), a.device)
    with _timed("colsum", 0.0, 2.0 * M * N):
        check(_fn("tad_colsum_bf16", op)(a.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), ws.numel(), M, N, _stream()), "tad_colsum_bf16")
    return out


# ----------------------------------------------------------------------------- attention
def attn_fwd(qkv, B: int, N: int, H: int, scale: float, out_dtype=None, want_lse=True, want_lo=False, q_prescaled=False, drop_p: float = 0.0,
             seed: int = 0, d: int = 64):
    """qkv [B*N, 3*H*d] bf16 or f16 (packed [B,N,3,H,d], d = 64 or 80) -> out [B*N, H*d], lse [B,H,N] f32.  want_lo: returns
    (out, lse, out_lo) with out_lo = what the 16-bit rounding of out dropped (for attn_bwd's delta).  q_prescaled: the q third already
    carries scale * log2(e) (linear_fwd_qkv's q_prescale = q_prescale_of(scale))"""
    op = _req16(qkv, "attn.qkv")
    out_dtype = _out16(out_dtype, op)
    if qkv.numel() != B * N * 3 * H * d:
        raise _lib.TadError(f"attn_fwd: qkv has {qkv.numel()} elements, expected {B * N * 3 * H * d}")
    out = torch.empty((B * N, H * d), dtype=out_dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if want_lse else None
    lo = torch.empty_like(out) if (want_lo and out.dtype in OP16_DTYPES) else None
    with _timed("attn_fwd", 4.0 * B * H * N * N * d, 2.0 * (4 + (lo is not None)) * B * N * H * d):
        check(_fn("tad_attn_fwd", op)(qkv.data_ptr(), out.data_ptr(), _dt(out), _p(lo), _p(lse), B, N, H, int(d), float(scale), int(bool(q_prescaled)),
                                      float(drop_p), int(seed) & 0xffffffff, _stream()), "tad_attn_fwd")
    return (out, lse, lo) if want_lo else (out, lse)


def attn_tuning(**knobs):
    """Scheduling knobs of the attention kernels (include/tad_mi355x.h: tad_attn_tuning); timing only, never results."""
    for k, v in knobs.items():
        check(_lib.load().tad_attn_tuning(k.encode(), int(v)), f"tad_attn_tuning({k}={v})")


def attn_bwd(qkv, out, dout, lse, B: int, N: int, H: int, scale: float, out_lo=None, q_prescaled=False, drop_p: float = 0.0, seed: int = 0,
             d: int = 64):
    """dqkv; its q slot is the gradient of the PLAIN q whether or not the q of `qkv` is pre-scaled"""
    op = _req16(qkv, "attn_bwd.qkv")
    for t, n in ((out, "out"), (dout, "dout")) + (((out_lo, "out_lo"),) if out_lo is not None else ()):
        _req16(t, "attn_bwd." + n, like=op)
        if t.numel() != B * N * H * d:
            raise _lib.TadError(f"attn_bwd: {n} has {t.numel()} elements, expected {B * N * H * d}")
    _req(lse, torch.float32, "attn_bwd.lse")
    if qkv.numel() != B * N * 3 * H * d or lse.numel() != B * H * N:
        raise _lib.TadError("attn_bwd: qkv / lse element count mismatch")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((_lib.load().tad_attn_bwd_scratch_bytes(B, N, H) // 4,), dtype=torch.float32, device=qkv.device)  # -rowsum(dout*out), -lse/scale
    with _timed("attn_bwd", 8.0 * B * H * N * N * d, 2.0 * (8 + (out_lo is not None)) * B * N * H * d):
        check(_fn("tad_attn_bwd", op)(qkv.data_ptr(), out.data_ptr(), _p(out_lo), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(),
                                       B, N, H, int(d), float(scale), int(bool(q_prescaled)), float(drop_p), int(seed) & 0xffffffff, _stream()),
              "tad_attn_bwd")
    return dqkv


# ----------------------------------------------------------------------------- mean-pool
def meanpool_fwd(x):
    _req(x, torch.float32, "meanpool.x")
    B, N, D = x.shape
    y = torch.empty((B, D), dtype=torch.float32, device=x.device)
    ws = workspace(B * _lib.POOL_SPLIT * D * 4, x.device)
    check(_lib.load().tad_meanpool_fwd(x.data_ptr(), y.data_ptr(), ws.data_ptr(), B, N, D, _stream()), "tad_meanpool_fwd")
    return y


def meanpool_bwd(dy, N: int, want_bf16=False):
    _req(dy, torch.float32, "meanpool_bwd.dy")
    B, D = dy.shape
    dx = torch.empty((B, N, D), dtype=torch.float32, device=dy.device)
    dxb = torch.empty((B, N, D), dtype=_op16, device=dy.device) if want_bf16 else None
    check(_fn("tad_meanpool_bwd", _op16)(dy.data_ptr(), dx.data_ptr(), _p(dxb), B, N, D, _stream()), "tad_meanpool_bwd")
    return dx, dxb


def sumsq(x, out):
    _req(x, torch.float32, "sumsq.x")
    lib = _lib.load()
    ws = workspace(lib.tad_sumsq_workspace_bytes(), x.device)
    check(lib.tad_sumsq_f32(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "tad_sumsq_f32")
    return out


def grad_norm_coef(x, inv_scale: float, max_norm: float = 0.0, out=None):
    """[norm of x with the loss scale removed, grad_scale coefficient for adamw_step (0 = skip: overflow), found-inf flag] as one f32[3]
    device tensor (tad_grad_norm_coef); nothing is read back"""
    _req(x, torch.float32, "grad_norm_coef.x")
    lib = _lib.load()
    if out is None:
        out = torch.empty(3, dtype=torch.float32, device=x.device)
    ws = workspace(lib.tad_sumsq_workspace_bytes(), x.device)
    with _timed("sumsq", 0.0, 4.0 * x.numel()):
        check(lib.tad_grad_norm_coef(x.data_ptr(), x.numel(), float(inv_scale), float(max_norm or 0.0), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                     _stream()), "tad_grad_norm_coef")
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, chunk_group, group_lr, group_wd, group_step, beta1, beta2, eps, param_bf16=None,
               grad_scale=None, sumsq_partials=None):
    """Fused AdamW over flat f32 buffers (tad_adamw_step); group_lr / group_wd / group_step are host sequences, one entry per
    parameter group (group_step: 1-based update count of the group's tensors after this call)."""
    import ctypes as C
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, torch.float32, "adamw." + nm)
    n = param.numel()
    if not (grad.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n):
        raise _lib.TadError("adamw_step: flat buffers differ in length")
    chunks = (n + _lib.ADAMW_CHUNK - 1) // _lib.ADAMW_CHUNK
    if chunk_group.dtype != torch.uint8 or chunk_group.numel() != chunks or not chunk_group.is_cuda:
        raise _lib.TadError(f"adamw_step: chunk_group must be a uint8 device tensor with {chunks} entries")
    op = torch.bfloat16
    if param_bf16 is not None:
        op = _req16(param_bf16, "adamw.param_bf16")
        assert param_bf16.numel() == n
    if sumsq_partials is not None:
        _req(sumsq_partials, torch.float32, "adamw.sumsq_partials")
        assert sumsq_partials.numel() == chunks
    if grad_scale is not None:
        _req(grad_scale, torch.float32, "adamw.grad_scale")
    ng = len(group_lr)
    assert len(group_wd) == ng and len(group_step) == ng
    lr = (C.c_float * ng)(*[float(v) for v in group_lr])
    wd = (C.c_float * ng)(*[float(v) for v in group_wd])
    st = (C.c_int32 * ng)(*[int(v) for v in group_step])
    with _timed("adamw", 0.0, 30.0 * n):
        check(_fn("tad_adamw_step", op)(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), _p(param_bf16),
                                         chunk_group.data_ptr(), n, lr, wd, ng, st, float(beta1), float(beta2), float(eps),
                                         _p(grad_scale), _p(sumsq_partials), _stream()), "tad_adamw_step")


# ----------------------------------------------------------------------------- MAE pre-training path
def _idx(t, name):
    if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
        raise _lib.TadError(f"{name}: expected a contiguous int32 GPU tensor")
    return t


def gather_rows(src, idx, bound_check=True):
    """out[r] = src[idx[r]]; src [R,D] f32, idx [n] int32"""
    _req(src, torch.float32, "gather_rows.src")
    _idx(idx, "gather_rows.idx")
    R, D = src.shape
    out = torch.empty((idx.numel(), D), dtype=torch.float32, device=src.device)
    with _timed("gather", 0.0, 8.0 * idx.numel() * D):
        check(_lib.load().tad_gather_rows_f32(src.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), D, _stream()), "tad_gather_rows_f32")
    return out


def scatter_rows(src, idx, n_rows):
    """out [n_rows, D] zeros except out[idx[r]] = src[r] (unique indices)"""
    _req(src, torch.float32, "scatter_rows.src")
    _idx(idx, "scatter_rows.idx")
    n, D = src.shape
    assert idx.numel() == n
    out = torch.zeros((n_rows, D), dtype=torch.float32, device=src.device)
    with _timed("gather", 0.0, 8.0 * n * D):
        check(_lib.load().tad_scatter_rows_f32(src.data_ptr(), idx.data_ptr(), out.data_ptr(), n, D, _stream()), "tad_scatter_rows_f32")
    return out


def mae_assemble(x_vis, mask_token, pos, vis_idx, mask_idx, B):
    """decoder input [B, Nv+Nm, D] = cat(x_vis + pos[vis], mask_token + pos[masked])"""
    _req(x_vis, torch.float32, "mae_assemble.x_vis")
    _req(mask_token, torch.float32, "mae_assemble.mask_token")
    _req(pos, torch.float32, "mae_assemble.pos")
    _idx(vis_idx, "mae_assemble.vis_idx")
    _idx(mask_idx, "mae_assemble.mask_idx")
    D = x_vis.shape[-1]
    Nv, Nm = vis_idx.numel() // B, mask_idx.numel() // B
    if x_vis.numel() != B * Nv * D or mask_token.numel() != D or pos.shape[-1] != D or pos.numel() // D < Nv + Nm:
        raise _lib.TadError("mae_assemble: inconsistent shapes")
    out = torch.empty((B, Nv + Nm, D), dtype=torch.float32, device=x_vis.device)
    with _timed("gather", 0.0, 12.0 * out.numel()):
        check(_lib.load().tad_mae_assemble(x_vis.data_ptr(), mask_token.data_ptr(), pos.data_ptr(), vis_idx.data_ptr(), mask_idx.data_ptr(),
                                           out.data_ptr(), B, Nv, Nm, D, _stream()), "tad_mae_assemble")
    return out


def mae_target(videos, mask_idx, tubelet, patch, mean, std, normalize_target=True):
    """videos [B,3,T,H,W] f32 (normalised) -> labels [B, Nm, tub*p*p*3] f32 for the masked tokens (engine_for_pretraining.py:51-66)"""
    import ctypes as C
    _req(videos, torch.float32, "mae_target.videos")
    _idx(mask_idx, "mae_target.mask_idx")
    B, Cc, T, H, W = videos.shape
    if Cc != 3:
        raise _lib.TadError("mae_target: clips must have 3 channels")
    Nm = mask_idx.numel() // B
    labels = torch.empty((B, Nm, tubelet * patch * patch * 3), dtype=torch.float32, device=videos.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    with _timed("mae_target", 0.0, 8.0 * labels.numel()):
        check(_lib.load().tad_mae_target(videos.data_ptr(), mask_idx.data_ptr(), labels.data_ptr(), B, Nm, T, H, W, tubelet, patch, m, s,
                                         int(bool(normalize_target)), _stream()), "tad_mae_target")
    return labels


def mse_loss(pred, target, want_grad=True):
    """returns (loss scalar tensor, grad|None) of nn.MSELoss()(pred, target)"""
    _req(pred, torch.float32, "mse.pred")
    _req(target, torch.float32, "mse.target")
    n = pred.numel()
    if target.numel() != n:
        raise _lib.TadError("mse_loss: pred and target differ in size")
    lib = _lib.load()
    partials = torch.empty(lib.tad_mse_loss_blocks(n), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    with _timed("mse", 0.0, (12.0 if want_grad else 8.0) * n):
        check(lib.tad_mse_loss(pred.data_ptr(), target.data_ptr(), n, partials.data_ptr(), _p(grad), _stream()), "tad_mse_loss")
    return partials.sum() / n, grad


def device_info():
    import ctypes as C
    cu, clk, ldsb = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    check(_lib.load().tad_device_info(C.byref(cu), C.byref(clk), C.byref(ldsb), name, 64), "tad_device_info")
    return {"cu_count": cu.value, "clock_khz": clk.value, "lds_bytes_per_cu": ldsb.value, "arch": name.value.decode()}


# ----------------------------------------------------------------------------- precise mode (parity gate)
def split_bf16x3(x, role_b: bool, stack: bool = False, dtype=torch.bfloat16):
    """x [M,K] f32 -> 16-bit [M,3K] ([hi|hi|lo] or [hi|lo|hi]) or, stacked, [3M,K]; hi = op16(x), lo = op16(x - hi)"""
    _req(x, torch.float32, "split.x")
    M, K = x.shape
    out = torch.empty((3 * M, K) if stack else (M, 3 * K), dtype=dtype, device=x.device)
    check(_fn("tad_split_bf16x3", dtype)(x.data_ptr(), out.data_ptr(), M, K, int(role_b), int(stack), _stream()), "tad_split_bf16x3")
    return out


def im2col_tubelets_f32(x, tubelet: int, patch: int):
    _req(x, torch.float32, "im2col_f32.x")
    B, Cc, T, H, W = x.shape
    ntok = (T // tubelet) * (H // patch) * (W // patch)
    cols = torch.empty((B * ntok, patch_embed_ldk(Cc, tubelet, patch)), dtype=torch.float32, device=x.device)
    check(_lib.load().tad_im2col_tubelets_f32(x.data_ptr(), cols.data_ptr(), B, Cc, T, H, W, tubelet, patch, _stream()),
          "tad_im2col_tubelets_f32")
    return cols


def attn_fwd_f32(qkv, B: int, N: int, H: int, scale: float, want_lse=False, d: int = 64, drop_p: float = 0.0, seed: int = 0):
    """f32 attention on the matrix pipe (precise mode; head dims without a 16-bit kernel: d = 80; attention dropout: drop_p, seed)"""
    _req(qkv, torch.float32, "attn_f32.qkv")
    if qkv.numel() != B * N * 3 * H * d:
        raise _lib.TadError("attn_fwd_f32: qkv element count mismatch")
    out = torch.empty((B * N, H * d), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if want_lse else None
    with _timed("attn_f32", 4.0 * B * H * N * N * d, 4.0 * 4 * B * N * H * d):
        check(_lib.load().tad_attn_fwd_f32(qkv.data_ptr(), out.data_ptr(), _p(lse), B, N, H, int(d), float(scale), float(drop_p),
                                           int(seed) & 0xffffffff, _stream()), "tad_attn_fwd_f32")
    return out, lse


def attn_bwd_f32(qkv, out, dout, lse, B: int, N: int, H: int, scale: float, d: int = 64, drop_p: float = 0.0, seed: int = 0):
    for t, n in ((qkv, "qkv"), (out, "out"), (dout, "dout"), (lse, "lse")):
        _req(t, torch.float32, "attn_bwd_f32." + n)
    if qkv.numel() != B * N * 3 * H * d or out.numel() != B * N * H * d or dout.numel() != out.numel():
        raise _lib.TadError("attn_bwd_f32: element count mismatch")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    with _timed("attn_f32", 8.0 * B * H * N * N * d, 4.0 * 8 * B * N * H * d):
        check(_lib.load().tad_attn_bwd_f32(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(),
                                           B, N, H, int(d), float(scale), float(drop_p), int(seed) & 0xffffffff, _stream()), "tad_attn_bwd_f32")
    return dqkv


def gelu_f32(h):
    _req(h, torch.float32, "gelu_f32.h")
    a = torch.empty_like(h)
    check(_lib.load().tad_gelu_f32(h.data_ptr(), a.data_ptr(), h.numel(), _stream()), "tad_gelu_f32")
    return a


def gelu_bwd_f32(dy, h):
    _req(dy, torch.float32, "gelu_bwd_f32.dy")
    _req(h, torch.float32, "gelu_bwd_f32.h")
    dh = torch.empty_like(h)
    check(_lib.load().tad_gelu_bwd_f32(dy.data_ptr(), h.data_ptr(), dh.data_ptr(), h.numel(), _stream()), "tad_gelu_bwd_f32")
    return dh


def colsum_window_f32(a, r0, rc):
    """a [B, R, N] f32 -> [N]: sum over the batch and rows r0 .. r0+rc-1 (no copy of the window)"""
    _req(a, torch.float32, "colsum_window.a")
    B, R, N = a.shape
    out = torch.empty(N, dtype=torch.float32, device=a.device)
    lib = _lib.load()
    ws = workspace(lib.tad_colsum_workspace_bytes(B * rc, N), a.device)
    check(lib.tad_colsum_window_f32(a.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), ws.numel(), B, R, N, int(r0), int(rc), _stream()),
          "tad_colsum_window_f32")
    return out


def colsum_f32(a):
    _req(a, torch.float32, "colsum_f32.a")
    M, N = a.shape
    out = torch.empty(N, dtype=torch.float32, device=a.device)
    check(_lib.load().tad_colsum_f32(a.data_ptr(), out.data_ptr(), M, N, _stream()), "tad_colsum_f32")
    return out
