#!/usr/bin/env python3
"""Compile one csrc/*.hip file to gfx950 assembly with the library's flags and print, per kernel, the register budget and a compact
run-length digest of the instructions that decide whether a tile loop is pipelined: LDS-DMA issues, LDS reads, MFMAs, waits, barriers,
ordinary global loads and scratch traffic.  This is how the compiler-inserted `s_waitcnt vmcnt(0)` in front of builtin transposed LDS
reads, the per-row global load in the residual epilogue and the spill reloads inside K loops were found (docs/DESIGN_HISTORY.md 3.2 / 3.4).

    python tools/isa_digest.py attn_bwd [--kernel dkv] [--width 160]
    python tools/isa_digest.py gemm --kernel "gemm_nt_kernelILi256ELi256ELi2ELi4ELi2ELi64ELi1ELi2ELb0ELb1ELb0ELb0E"

Needs hipcc only (no GPU)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import build as B  # noqa: E402

KEEP = re.compile(r"^\s+(s_waitcnt|s_barrier|buffer_load|buffer_store|global_load|global_store|ds_read|ds_write|ds_bpermute|v_mfma|scratch_|s_setprio)")


def digest(lines, width):
    out, last, n = [], None, 0
    for ln in lines:
        if not KEEP.match(ln):
            continue
        tok = ln.split()
        k = tok[0]
        if k.startswith("buffer_load") and "lds" in ln:
            k = "DMA"
        elif k.startswith("v_mfma"):
            k = "mfma"
        elif k.startswith("ds_read"):
            k = "lds_rd" + ("_tr" if "_tr_" in k else "")
        elif k.startswith("ds_write"):
            k = "lds_wr"
        elif k == "s_waitcnt":
            k = "wait " + " ".join(t for t in tok[1:] if "cnt" in t)
        if k == last:
            n += 1
        else:
            if last is not None:
                out.append(f"{n}x {last}" if n > 1 else last)
            last, n = k, 1
    if last is not None:
        out.append(f"{n}x {last}" if n > 1 else last)
    text, line = [], ""
    for item in out:
        if len(line) + len(item) + 3 > width:
            text.append(line)
            line = ""
        line += (" | " if line else "") + item
    text.append(line)
    return "\n".join(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="csrc file name without extension (gemm, attn_fwd, attn_bwd, layernorm, ...)")
    ap.add_argument("--kernel", default="", help="substring of the (mangled) kernel name; default: every kernel, registers only")
    ap.add_argument("--width", type=int, default=160)
    a = ap.parse_args()
    src = os.path.join(ROOT, "simple_tad_amd", "csrc", a.source + ".hip")
    flags = [f for f in B.FLAGS if f != "-fPIC"]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, a.source + ".s")
        subprocess.run([B._hipcc(), *flags, "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), src, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    # kernel bodies: "<name>:" ... "s_endpgm"; the resource comment block follows each body
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):", text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i
        while j < len(text) and not text[j].startswith(".Lfunc_end"):  # (a body can hold several s_endpgm)
            j += 1
        res = {}
        k = j
        while k < len(text) and not re.match(r"^_Z\w+:", text[k]):
            mm = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", text[k])
            if mm:
                res.setdefault(mm.group(1), int(mm.group(2)))
            if "Occupancy" in res:
                break
            k += 1
        if not a.kernel or a.kernel in name:
            flag = "  <-- SPILLS" if res.get("ScratchSize", 0) else ""
            print(f"{name}\n    vgpr {res.get('NumVgprs')} agpr {res.get('NumAgprs')} scratch {res.get('ScratchSize')} B lds {res.get('LDSByteSize')} B "
                  f"occupancy {res.get('Occupancy')}{flag}")
            if a.kernel:
                print(digest(text[i:j], a.width))
        i = j + 1


if __name__ == "__main__":
    main()
