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
    assert len(_lib.F16_TWINS) == 22
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
