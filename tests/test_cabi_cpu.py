"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/tad_mi355x.h declares, the ctypes binding covers the same set, and host-side argument
validation fails loudly (no compute launches here -- there is no GPU in this container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tad_mi355x.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tad_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib_path():
    from simple_tad_amd import build
    return build.build(verbose=False)


def test_library_loads_and_exports_header_symbols(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in tad_mi355x.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (tad_[a-z0-9_]+)", out))
    assert exported == set(syms), exported ^ set(syms)


def test_ctypes_binding_matches_header(lib_path):
    from simple_tad_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    assert lib.tad_abi_version() == _lib.ABI_VERSION == 4


def test_every_operand_entry_point_has_a_half_twin(lib_path):
    """the IEEE-half twins (tad_*_f16) mirror their bf16 entry points one-to-one: exported, bound with the same signature"""
    from simple_tad_amd import _lib
    lib = _lib.load()
    assert len(_lib.F16_TWINS) == 23
    for bf, half in _lib.F16_TWINS.items():
        assert half.endswith("f16") or "f16x3" in half
        assert _lib.SIGNATURES[bf] == _lib.SIGNATURES[half]
        assert hasattr(lib, bf) and hasattr(lib, half)
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    # the twin takes TAD_F16 (2) where its sibling takes TAD_BF16 (1) as the 16-bit output type, and refuses the other one
    assert lib.tad_linear_fwd_f16(p, p, None, p, 1, 0, None, None, None, None, 1, None, 0, 16, 16, 64, None) == -1
    assert b"y_dtype" in lib.tad_last_error_string()
    assert lib.tad_linear_fwd(p, p, None, p, 2, 0, None, None, None, None, 1, None, 0, 16, 16, 64, None) == -1
    assert b"y_dtype" in lib.tad_last_error_string()


def test_host_validation_without_gpu(lib_path):
    """argument checks run before any launch, so they are testable on CPU"""
    from simple_tad_amd import _lib
    lib = _lib.load()
    assert lib.tad_attn_fwd(None, None, 1, None, None, 1, 1, 1, 64, 0.125, 0, 0.0, 0, None) == -1
    assert b"null" in lib.tad_last_error_string()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.tad_attn_fwd(p, p, 1, None, None, 1, 8, 1, 32, 0.125, 0, 0.0, 0, None) == -1  # head_dim != 64
    assert b"head_dim" in lib.tad_last_error_string()
    assert lib.tad_linear_fwd(p, p, None, p, 1, 0, None, None, None, None, 1, None, 0, 16, 16, 60, None) == -1  # K % 64
    assert b"K=60" in lib.tad_last_error_string()
    assert lib.tad_layernorm_fwd(p, p, p, p, 1, None, None, 4, 6, 1e-6, None) == -1  # D % 4
    assert lib.tad_im2col_tubelets(p, p, 1, 3, 3, 16, 16, 2, 8, None) == -1  # T % tubelet
    lr = (ctypes.c_float * 2)(1e-3, 1e-3)
    wd = (ctypes.c_float * 2)(0.05, 0.0)
    st = (ctypes.c_int32 * 2)(1, 0)
    assert lib.tad_adamw_step(p, p, p, p, None, p, 4096, lr, wd, 2, st, 0.9, 0.999, 1e-8, None, None, None) == -1  # steps count from 1
    assert b"step" in lib.tad_last_error_string()
    st[1] = 1
    assert lib.tad_adamw_step(p, p, p, p, None, p, 4098, lr, wd, 2, st, 0.9, 0.999, 1e-8, None, None, None) == -1  # n % 4
    assert lib.tad_adamw_step(p, p, p, p, None, p, 4096, lr, wd, 200, st, 0.9, 0.999, 1e-8, None, None, None) == -1  # > MAX_GROUPS


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from simple_tad_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TadError, match="no CPU fallback"):
        _lib.load()


def test_linear_workspace_bytes_is_the_split_plan_of_the_launcher(lib_path):
    """tad_linear_workspace_bytes and the launcher share ONE copy of the split plan's geometry.  Known answers on the 256 CUs
    the library assumes without a device: ViT-B's Linears never split along K under the default knobs; ViT-L's N = 1024, K = 4096 Linears at
    32 clips leave a 16-tile tail behind 3 whole rounds = 16 tiles x 8 shares of 256 KiB + the 4 KiB header; and a problem taller than the
    32-bit offset limit is sized by the row ranges the launcher cuts it into."""
    from simple_tad_amd import _lib
    lib = _lib.load()
    M = 32 * 1568
    for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
        assert lib.tad_linear_workspace_bytes(M, n, k) == 0, (n, k)
    assert lib.tad_linear_workspace_bytes(M, 1024, 4096) == 4096 + 16 * 8 * 256 * 256 * 4
    # forced split-K ("splitk_tail" 2) takes ViT-B's 76-tile tails as three shares each
    v = ctypes.c_int(-1)
    assert lib.tad_linear_tuning_get(b"splitk_tail", ctypes.byref(v)) == 0 and v.value == 1
    assert lib.tad_linear_tuning(b"splitk_tail", 2) == 0
    try:
        assert lib.tad_linear_tuning_get(b"splitk_tail", ctypes.byref(v)) == 0 and v.value == 2
        assert lib.tad_linear_workspace_bytes(M, 768, 3072) == 4096 + 78 * 3 * 256 * 256 * 4
    finally:
        assert lib.tad_linear_tuning(b"splitk_tail", 1) == 0
    # 573 952 rows x 1024 columns exceed the 32-bit offsets of an f32 epilogue: the launcher runs 523 776 + 50 176 rows (f32) or 524 032 +
    # 49 920 (16-bit), and the query sizes the workspace for THOSE launches (the 16-tile tail of the 50 176-row range: 128 partial tiles),
    # not for the 8-tile tail the whole problem would have (64 partial tiles: too small, the split used to drop out quietly)
    assert lib.tad_linear_workspace_bytes(523776 + M, 1024, 4096) == 4096 + 16 * 8 * 256 * 256 * 4
    assert lib.tad_linear_tuning_get(b"no_such_knob", ctypes.byref(v)) == -1 and b"unknown key" in lib.tad_last_error_string()


def test_tuning_scope_restores_what_it_replaced(lib_path):
    """two models in one process do not see each other's plan: a TuningScope puts back the knobs, the precision mode and the q pre-scale
    contract it found, also when the body raises; scopes nest; DataParallel's plan is such a scope"""
    import simple_tad_amd as T
    from simple_tad_amd import kernels as K, ops
    from simple_tad_amd.tuning import TuningScope
    base = {k: K.linear_tuning_get(k) for k in K.LINEAR_TUNING_DEFAULTS}
    assert base == K.LINEAR_TUNING_DEFAULTS, "the library's initial knobs are the documented defaults"
    with TuningScope(precision="half", attn_q_prescale=False, persistent=0, group_m=4):
        assert (K.linear_tuning_get("persistent"), K.linear_tuning_get("group_m")) == (0, 4)
        assert T.get_precision() == "half" and ops.get_attn_q_prescale() is False
        with TuningScope(persistent=1, variant=7):
            assert (K.linear_tuning_get("persistent"), K.linear_tuning_get("variant"), K.linear_tuning_get("group_m")) == (1, 7, 4)
        assert (K.linear_tuning_get("persistent"), K.linear_tuning_get("variant")) == (0, 0)
    assert {k: K.linear_tuning_get(k) for k in K.LINEAR_TUNING_DEFAULTS} == base
    assert T.get_precision() == "fast" and ops.get_attn_q_prescale() is True
    with pytest.raises(RuntimeError, match="boom"):
        with TuningScope(tail_192=0, precision="half"):
            raise RuntimeError("boom")
    assert K.linear_tuning_get("tail_192") == 1 and T.get_precision() == "fast"
    with pytest.raises(Exception):  # a refused value leaves nothing half-applied
        with TuningScope(group_m=2, variant=6):
            pass
    assert K.linear_tuning_get("group_m") == 0 and K.linear_tuning_get("variant") == 0
    with pytest.raises(ValueError, match="unknown"):
        TuningScope(no_such_knob=1)
    # a second thread's scope waits for the first one's instead of interleaving two plans
    import threading
    seen = []
    def other():
        with TuningScope(group_m=16):
            seen.append(K.linear_tuning_get("group_m"))
    with TuningScope(group_m=2):
        th = threading.Thread(target=other)
        th.start()
        th.join(0.2)
        assert th.is_alive() and seen == [] and K.linear_tuning_get("group_m") == 2
    th.join(5)
    assert seen == [16] and K.linear_tuning_get("group_m") == 0
