#!/usr/bin/env python
"""Run a script of this repo against a VARIANT library (tools/mk_variant.py) instead of the product build:

    python tools/run_with_lib.py tools/ab/lib_NAME.so bench.py --workload ncd-incre --no-cpu-baseline
    python tools/run_with_lib.py tools/ab/lib_NAME.so -m pytest tests -m gpu -q -k regul

For whole-loop measurements (frames/s of ncd-incre, Tier A iteration times) that tools/ab_build.py's kernel timing does
not cover.  Measurement aid only: the product always loads shine_mapping_amd/lib/libshine_hip.so."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shine_mapping_amd import _lib  # noqa: E402

lib_path, script = os.path.abspath(sys.argv[1]), sys.argv[2]
if not os.path.isfile(lib_path):
    raise SystemExit("no such library: %s" % lib_path)
_lib.LIB_PATH = lib_path
_lib.CHECK_LIB_PATH = lib_path  # a variant has the check library's composition (tools/mk_variant.py)
_lib._lib = None
_lib._check = None
print("[run_with_lib] %s" % lib_path, file=sys.stderr)
if script == "-m":  # python tools/run_with_lib.py LIB -m pytest tests -m gpu -k regul
    sys.argv = sys.argv[3:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    sys.argv = [script] + sys.argv[3:]
    runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")
