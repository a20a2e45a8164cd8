#!/usr/bin/env python
"""Build a VARIANT of libshine_hip.so for an in-process A/B on the GPU box (tools/ab_build.py):

    python tools/mk_variant.py NAME [-DMACRO=1 ...] [shine_step_v3.hip ... | all]

recompiles the named sources (default: shine_step_v3.hip) with the extra flags,
links them with the other objects of the current build (shine_mapping_amd/build/*.o) and writes tools/ab/lib_NAME.so
(git-ignored; it travels with the gpurun snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shine_mapping_amd import build as b  # noqa: E402

name = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
srcs = [a for a in sys.argv[2:] if a.endswith(".hip")] or ["shine_step_v3.hip"]
if "all" in sys.argv[2:]:  # a macro that changes a shared struct (the kernel argument block): every source
    srcs = list(b.sources()) + ["shine_step_v0.hip"]
b.build(verbose=False)
out_dir = os.environ.get("AB_DIR") or os.path.join(ROOT, "tools", "ab")  # (tools/ab/ does not travel with gpurun: .gpurunignore)
os.makedirs(out_dir, exist_ok=True)
# a variant has the composition of the CHECK library (product objects, shine_step_v0.hip in its training build), so
# kernel_variant 1 works against it too
jobs = [(os.path.join(b.CSRC, f), os.path.join(b.OBJDIR, f.replace(".hip", ".o")), [], f) for f in b.sources()]
jobs.append((os.path.join(b.CSRC, "shine_step_v0.hip"), os.path.join(b.OBJDIR, "check_shine_step_v0_train.o"), [],
             "shine_step_v0.hip"))
objs = []
procs = []
for path, obj, extra, fname in jobs:
    if fname in srcs:
        obj = os.path.join(out_dir, "%s_%s.o" % (name, fname.replace(".hip", "")))
        procs.append(subprocess.Popen([b.HIPCC] + b.FLAGS + extra + flags + ["-c", path, "-o", obj]))
    objs.append(obj)
for pr in procs:
    if pr.wait() != 0:
        sys.exit("hipcc failed")
lib = os.path.join(out_dir, "lib_%s.so" % name)
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
