"""Build libhallo_amd.so (the gfx950 operator library) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the authoring container as
well as on the MI355X box.  The library is placed next to this file so that it travels with
the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhallo_amd.so")
SOURCES = ["gemm.hip", "gemm3.hip", "gemm4.hip", "gemm_rs.hip", "gemm_rs2.hip", "gemm_ff.hip", "attention.hip", "attention40.hip", "fp8.hip", "fused_xattn.hip", "norm_elementwise.hip", "wav2vec.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_args.h"), os.path.join(CSRC, "attn_args.h"), os.path.join(HERE, "..", "include", "hallo_amd.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libhallo_amd.so.  Returns the library path."""
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
            if os.environ.get("HALLO_ABLATIONS") == "1":      # timing-ablation switches of the row-stationary GEMMs (tools/cbench)
                cmd.insert(1, "-DHALLO_ABLATIONS")
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
