"""Build libtad_mi355x.so (gfx950) in-tree with hipcc.  `python -m simple_tad_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# TAD_BUILD_LIB / TAD_BUILD_DEFINES: experiment builds next to the production library (own object directory), e.g.
#   TAD_BUILD_LIB=libtad_spread.so TAD_BUILD_DEFINES="-DTAD_DMA_SPREAD=1" python -m simple_tad_amd.build --force   -> build_exp/libtad_spread.so
# and TAD_LIB=<path> selects the library a process loads (_lib.py).
# TAD_BUILD_ABLATION=1 (the diagnostic build: -DTAD_GEMM_ABLATION, in-kernel stamps / clock readings / debug knobs) never writes the production
# library: without TAD_BUILD_LIB it builds build_exp/libtad_ablation.so in its own object directory.
_ABLATION = os.environ.get("TAD_BUILD_ABLATION") == "1"
_LIB_NAME = os.environ.get("TAD_BUILD_LIB", "libtad_ablation.so" if _ABLATION else "libtad_mi355x.so")
# The production library is the ONLY built file in the package directory; every experiment / ablation build (and its objects) goes to
# <repo>/build_exp/ -- git-ignored like every built artefact, but not gpurun-ignored, so it travels to the GPU box with the snapshot.
EXP_DIR = os.path.join(os.path.dirname(HERE), "build_exp")
_PRODUCTION = _LIB_NAME == "libtad_mi355x.so"
LIB = os.path.join(HERE if _PRODUCTION else EXP_DIR, os.path.basename(_LIB_NAME))
SOURCES = ["capi.hip", "elementwise.hip", "patch_embed.hip", "layernorm.hip", "gemm.hip", "gemm_w4.hip", "attn_fwd.hip", "attn_bwd.hip", "attn_f32.hip", "precise.hip", "optim.hip", "mae.hip", "metrics.hip", "collective.hip"]
# Sources that touch 16-bit GEMM / attention operands are compiled a second time with -DTAD_OPND_F16: the same kernels for IEEE half
# operands, exported as tad_*_f16 (csrc/common.h, csrc/opnd_f16_names.h; include/tad_mi355x.h "IEEE half operand twins").
F16_SOURCES = ["elementwise.hip", "patch_embed.hip", "layernorm.hip", "gemm.hip", "gemm_w4.hip", "attn_fwd.hip", "attn_bwd.hip", "optim.hip"]
# gemm_w4.hip (four waves of 128 x 128 outputs: 256 accumulator registers per lane) needs its accumulators in the AGPR half of the register
# file: compiled without the vgpr-form switch below.  It includes gemm.hip for the kernel template.
NO_VGPR_FORM = {"gemm_w4.hip"}
EXTRA_DEPS = {"gemm_w4.hip": ["gemm.hip"]}
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
    objdir = os.path.join(HERE, "build") if _PRODUCTION else os.path.join(EXP_DIR, "obj_" + os.path.splitext(os.path.basename(LIB))[0])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "opnd_f16_names.h"), os.path.join(os.path.dirname(HERE), "include", "tad_mi355x.h")]

    def compile_one(job):
        src, half = job
        obj = os.path.join(objdir, src.replace(".hip", "_f16.o" if half else ".o"))
        srcp = os.path.join(CSRC, src)
        deps = [srcp, *headers, *[os.path.join(CSRC, d) for d in EXTRA_DEPS.get(src, [])]]
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(p) for p in deps):
            return obj
        flags = [f for f in FLAGS if f not in ("-mllvm", "-amdgpu-mfma-vgpr-form=1")] if src in NO_VGPR_FORM else FLAGS
        cmd = [hipcc, *flags, *(["-DTAD_OPND_F16"] if half else []), *(["-DTAD_GEMM_ABLATION"] if _ABLATION else []),
               *os.environ.get("TAD_BUILD_DEFINES", "").split(), "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    # longest jobs first (gemm.hip takes ~35 s per pass); TAD_BUILD_JOBS bounds the parallelism (default: the CPU count, at most 8)
    jobs = sorted([(s, False) for s in SOURCES] + [(s, True) for s in F16_SOURCES], key=lambda j: -os.path.getsize(os.path.join(CSRC, j[0])))
    workers = int(os.environ.get("TAD_BUILD_JOBS", 0)) or min(8, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
