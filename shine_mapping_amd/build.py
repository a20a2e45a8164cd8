"""Build the HIP libraries for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m shine_mapping_amd.build [--force]

  lib/libshine_hip.so    the PRODUCT: csrc/*.hip — the fused step (shine_step_v3.hip) and everything around it
  lib/libshine_check.so  the product's objects + shine_step_v0.hip (the lane-per-point reference step): the on-device
                         cross-check of the GPU tests.  Only tests / tools load it (StepOptions.kernel_variant 1).
  lib/_shine_ext.so      csrc/shine_torch_ext.cpp: Tier A's autograd nodes as a torch C++ extension over the C ABI (host code
                         only; links libshine_hip.so through rpath $ORIGIN)

Both land in shine_mapping_amd/lib/ (git-ignored, but they travel with the gpurun snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libshine_hip.so")
CHECK_LIB = os.path.join(LIBDIR, "libshine_check.so")
EXT_LIB = os.path.join(LIBDIR, "_shine_ext.so")  # Tier A's autograd nodes in C++ (csrc/shine_torch_ext.cpp): host code, g++
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(os.path.dirname(HERE), "include")]


CHECK_ONLY = ("shine_step_v0.hip",)  # the lane-per-point reference step: tests / tools only


def sources():
    """the product library's translation units"""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in CHECK_ONLY)


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/shine_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    h.update(" ".join(f for f in FLAGS if not os.path.isabs(f)).encode())  # (not the checkout's own path: the snapshot on a GPU box
    #                                                                          lives elsewhere and must not rebuild for that)
    return h.hexdigest()


def _ext_digest(hip_digest: str) -> str:
    """the extension is host code against THIS interpreter's torch: its stamp also names the torch build and the C++ ABI flag, so
    a torch upgrade rebuilds it instead of leaving a library with undefined symbols in place"""
    try:
        import torch

        tag = "%s|abi%d" % (torch.__version__, int(torch._C._GLIBCXX_USE_CXX11_ABI))
    except Exception as e:  # (no torch: the extension cannot be built either)
        tag = "no-torch:%s" % type(e).__name__
    return hashlib.sha256((hip_digest + "|" + tag).encode()).hexdigest()


def _read(path):
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        return f.read()


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libshine_hip.stamp")
    ext_stamp = os.path.join(LIBDIR, "_shine_ext.stamp")  # its own stamp: "built", or "failed" for this very input (not retried)
    dig = _digest()
    ext_dig = _ext_digest(dig)
    hip_fresh = (not force and os.path.isfile(LIB) and os.path.isfile(CHECK_LIB) and _read(stamp) == dig)
    if hip_fresh:
        st = _read(ext_stamp)
        if (st == ext_dig and os.path.isfile(EXT_LIB)) or st == ext_dig + ":failed":
            return LIB
        _build_extension_optional(verbose, ext_stamp, ext_dig)
        return LIB
    if not os.path.isfile(HIPCC):
        raise RuntimeError("hipcc not found at %s; cannot build libshine_hip.so" % HIPCC)

    def compile_one(job):
        src_path, obj, extra = job
        cmd = [HIPCC] + FLAGS + extra + ["-c", src_path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src_path, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    product = [(os.path.join(CSRC, src), os.path.join(OBJDIR, src.replace(".hip", ".o")), []) for src in sources()]
    v0_train = (os.path.join(CSRC, "shine_step_v0.hip"), os.path.join(OBJDIR, "check_shine_step_v0_train.o"), [])
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, product + [v0_train]))
    n = len(product)
    product_objs, check_objs = objs[:n], objs[n:]

    def link(out, inputs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + inputs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))

    link(LIB, product_objs)
    # the check library: the product's objects + the lane-per-point reference step
    link(CHECK_LIB, product_objs + check_objs)
    _write(stamp, dig)  # the HIP libraries stand on their own: the extension below is optional
    _build_extension_optional(verbose, ext_stamp, ext_dig)
    return LIB


def _build_extension_optional(verbose, ext_stamp, ext_dig):
    """lib/_shine_ext.so is an accelerator of Tier A's host side, not a requirement: without a host g++ / matching torch headers
    the Python autograd nodes (autograd_ops.py) serve — warn, remember the failure for this input, carry on."""
    try:
        build_extension(verbose)
        _write(ext_stamp, ext_dig)
    except Exception as e:
        import warnings

        if os.path.isfile(EXT_LIB):
            os.remove(EXT_LIB)  # (a stale library from another torch would fail at import with undefined symbols)
        _write(ext_stamp, ext_dig + ":failed")
        warnings.warn("shine_mapping_amd: lib/_shine_ext.so (Tier A's C++ autograd nodes) was not built — the Python nodes are used "
                      "instead (same launches, more host time per iteration): %s" % str(e)[:2000])


def build_extension(verbose: bool = True) -> str:
    """lib/_shine_ext.so: csrc/shine_torch_ext.cpp against this interpreter's torch (host code only: plain g++)."""
    import sysconfig

    import torch

    ti = os.path.dirname(os.path.abspath(torch.__file__))
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_shine_ext", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ti, "include"), "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"],  # (pybind11: the headers torch bundles under its include/)
           os.path.join(CSRC, "shine_torch_ext.cpp"), "-o", EXT_LIB,
           "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
           "-L" + LIBDIR, "-lshine_hip", "-Wl,-rpath," + os.path.join(ti, "lib"), "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building _shine_ext.so failed:\n%s\n%s" % (r.stdout, r.stderr[-4000:]))
    return EXT_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
