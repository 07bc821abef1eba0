"""In-tree build of libnd4hip.so (hipcc, gfx950 only) and of the optional N-API shim.

`python -m nd4js_amd.build` or `__graft_entry__.build()`. Objects are cached by mtime under
nd4js_amd/csrc/_obj/ so an edit of one kernel file recompiles one file. hipcc cross-compiles
without a GPU; the built .so travels to the GPU box with the tree (git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libnd4hip.so")
ADDON = os.path.join(HERE, "js", "nd4hip_napi.node")
ARCH = "gfx950"
HIP_SOURCES = ["nd4hip_core.hip", "nd4hip_api.hip", "nd4hip_host.hip", "gemm.hip", "lu.hip", "qr.hip", "svd.hip", "svd_block.hip", "trsm.hip", "chol.hip", "hess.hip", "bidiag.hip"]
# -pragma-unroll-threshold: the register-resident panel kernels fully unroll 16 columns x R rows x 16 FMAs;
# below the default threshold LLVM silently keeps a rolled loop and the row arrays fall into scratch memory.
HIPCC_FLAGS = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-ffp-contract=on",
               "-mllvm", "-pragma-unroll-threshold=200000", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "nd4hip.h"))
    jobs = []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer(obj, [src] + headers):
            jobs.append([hipcc] + HIPCC_FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in HIP_SOURCES]
    if force or jobs or _newer(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
    return LIB


def build_addon(force=False, verbose=False):
    """N-API shim (plain C, no node-gyp): only if node headers are present."""
    inc = "/usr/include/node"
    src = os.path.join(CSRC, "napi_shim.c")
    if not (os.path.exists(os.path.join(inc, "node_api.h")) and os.path.exists(src)):
        return None
    if force or _newer(ADDON, [src, os.path.join(HERE, "..", "include", "nd4hip.h")]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", "-I" + inc, "-I" + os.path.join(HERE, "..", "include"),
               "-o", ADDON, src, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return ADDON


def build_all(force=False, verbose=False):
    return build_lib(force, verbose), build_addon(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
