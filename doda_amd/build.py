"""Build libdoda_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

`python -m doda_amd.build` or `doda_amd.build.build_native()`.  hipcc cross-compiles without a
GPU, so this also runs in the CPU-only build container; the resulting .so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdoda_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in ordinary VGPRs.  With the default (AGPR) form
# hipcc rotated the accumulators through 8 v_accvgpr_mov per unit of the conv ring (ISA check); the
# kernels use 40-150 VGPRs, so the unified register file has room.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"] + os.environ.get("DODA_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src):
    h = hashlib.sha1()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))   # every shared header
    for p in [src] + headers + [os.path.join(HERE, "..", "include", "doda_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    stamp_file = obj + ".sha1"
    stamp = _stamp(src)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj, True


def build_native(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link doda_amd/libdoda_hip.so.  Returns the path."""
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[doda_amd.build] linked", LIB)
    elif verbose:
        print("[doda_amd.build] up to date:", LIB)
    return LIB


EXT_SRC = os.path.join(HERE, "csrc_ext", "doda_torch.cpp")
EXT_LIB = os.path.join(HERE, "_doda_torch.so")


def build_torch_ext(force=False, verbose=True):
    """g++ the thin PyTorch-ROCm glue (host-only C++) against torch + libdoda_hip.so, in-tree."""
    import sysconfig
    import torch
    if not os.path.exists(LIB):
        build_native(verbose=verbose)
    stamp_file = EXT_LIB + ".sha1"
    h = hashlib.sha1()
    for p in (EXT_SRC, os.path.join(HERE, "..", "include", "doda_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    stamp = h.hexdigest()
    if (not force and os.path.exists(EXT_LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        if verbose:
            print("[doda_amd.build] up to date:", EXT_LIB)
        return EXT_LIB
    ti = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_doda_torch",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi,
           "-I" + os.path.join(ti, "include"), "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"],
           EXT_SRC, "-o", EXT_LIB,
           "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
           "-ltorch_python", "-L" + HERE, "-l:libdoda_hip.so",
           "-Wl,-rpath," + os.path.join(ti, "lib"), "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch extension build failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-6000:]))
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print("[doda_amd.build] linked", EXT_LIB)
    return EXT_LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    build_torch_ext(force="--force" in sys.argv)
