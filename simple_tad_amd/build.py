"""Build libtad_mi355x.so (gfx950) in-tree with hipcc.  `python -m simple_tad_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# TAD_BUILD_LIB / TAD_BUILD_DEFINES: experiment builds next to the production library (own object directory), e.g.
#   TAD_BUILD_LIB=libtad_spread.so TAD_BUILD_DEFINES="-DTAD_DMA_SPREAD=1" python -m simple_tad_amd.build --force
# and TAD_LIB=<path> selects the library a process loads (_lib.py).
LIB = os.path.join(HERE, os.environ.get("TAD_BUILD_LIB", "libtad_mi355x.so"))
SOURCES = ["capi.hip", "elementwise.hip", "layernorm.hip", "gemm.hip", "attn_fwd.hip", "attn_bwd.hip", "precise.hip", "optim.hip", "mae.hip", "metrics.hip", "collective.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result",
         # keep MFMA accumulators in the (unified) VGPR file: without it the compiler parks them in AGPRs and pays a
         # v_accvgpr_read/write per element around every softmax / epilogue
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    objdir = os.path.join(HERE, "build" if "TAD_BUILD_LIB" not in os.environ else "build_" + os.path.splitext(os.path.basename(LIB))[0])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), os.path.getmtime(os.path.join(CSRC, "common.h")),
                                                                          os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "tad_mi355x.h")))):
            return obj
        cmd = [hipcc, *FLAGS, *(["-DTAD_GEMM_ABLATION"] if os.environ.get("TAD_BUILD_ABLATION") == "1" else []),
               *os.environ.get("TAD_BUILD_DEFINES", "").split(), "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
