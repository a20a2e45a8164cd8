"""CPU suite: the C-ABI library is built, loads, and exports every symbol include/shine_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "shine_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shine_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "shine_train_step" in syms and "shine_forward" in syms and len(syms) >= 8


def test_library_builds_and_exports_every_declared_symbol():
    from shine_mapping_amd import build

    path = build.build(verbose=False)
    assert os.path.isfile(path)
    h = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(h, name), "libshine_hip.so does not export %s" % name


def test_binding_covers_the_header():
    from shine_mapping_amd import _lib

    assert _lib.exported_symbols() == declared_symbols()
    lib = _lib.lib()
    assert lib.shine_version() >= 100
    assert lib.shine_error_string(0) == b"ok"


def test_host_side_argument_errors_do_not_need_a_gpu():
    from shine_mapping_amd import _lib

    lib = _lib.lib()
    out = ctypes.c_void_p()
    assert lib.shine_tables_create(0, ctypes.byref(out)) == -1  # SHINE_E_INVALID
    assert lib.shine_tables_create(3, ctypes.byref(out)) == 0
    cap, cnt = ctypes.c_int64(-1), ctypes.c_int64(-1)
    assert lib.shine_tables_stats(out, 0, ctypes.byref(cap), ctypes.byref(cnt)) == 0
    assert (cap.value, cnt.value) == (0, 0)
    assert lib.shine_tables_stats(out, 7, ctypes.byref(cap), ctypes.byref(cnt)) == -1
    assert b"slot" in lib.shine_error_string(-1)
    assert lib.shine_tables_destroy(out) == 0


def test_argument_checks_of_the_per_iteration_entry_points():
    """Bad arguments are rejected with SHINE_E_INVALID (-1) before anything touches a device; workspace-size queries are
    host arithmetic.  (The reference raises Python exceptions for the same misuse; the binding turns -1 into one.)"""
    import pytest

    from shine_mapping_amd import _lib

    lib = _lib.lib()
    C = ctypes
    need = C.c_size_t(0)
    # sampler: size query, then a slice that does not fit the draw / a pool that does not fit int32
    assert lib.shine_sample_sorted(1000, 4096, 1, 0, None, None, 0, None, None, None, C.byref(need), None) == 0
    assert need.value >= 5 * 8  # one fp64 block sum per 1024 draws (+ the closing spacing)
    assert lib.shine_sample_sorted_slice(1000, 4096, 4000, 200, 1, 0, None, None, None, 0, None, None, None, C.byref(need), None) == -1
    assert lib.shine_sample_sorted_slice(1000, 4096, -1, 10, 1, 0, None, None, None, 0, None, None, None, C.byref(need), None) == -1
    assert lib.shine_sample_sorted(1 << 40, 16, 1, 0, None, None, 0, None, None, None, C.byref(need), None) == -1
    assert lib.shine_sample_sorted(1000, 16, 1, 0, None, None, 0, None, None, None, None, None) == -1
    assert b"shine_sample_sorted" in lib.shine_last_error() if hasattr(lib, "shine_last_error") else True
    # importance sweep / regulariser: null handles and level counts
    assert lib.shine_importance_sweep(None, None, None, None, None, None, None, None, 0, None, None, None, None, 1, None, 0,
                                      None, 0, None) == -1
    with pytest.raises(_lib.ShineHipError):
        _lib.check(lib.shine_importance_sweep(None, None, None, None, None, None, None, None, 0, None, None, None, None, 1,
                                              None, 0, None, 0, None), "shine_importance_sweep")
    need = C.c_size_t()
    assert lib.shine_importance_chunks(None, 1000, 64, 2, None, None, 8, None, C.byref(need), None) == 0 and need.value > 0
    assert lib.shine_importance_chunks(None, 1000, 64, 2, None, None, 7, None, C.byref(need), None) == -1  # ceil(1000 / 128) = 8
    assert lib.shine_importance_chunks(None, 1000, 0, 2, None, None, 8, None, C.byref(need), None) == -1
    # the sweep's sizes: 64 chunks per launch at most, fewer under a scratch budget, never less than one
    rows3 = (C.c_int64 * 3)(1000, 5000, 20000)
    group, sb, wb = C.c_int32(), C.c_size_t(), C.c_size_t()
    per_chunk = sum((r + 1) * 32 + ((r + 1 + 15) & ~15) for r in rows3)
    assert lib.shine_importance_sweep_sizes(3, rows3, 150, 4096, 0, C.byref(group), C.byref(sb), C.byref(wb)) == 0
    assert group.value == 64 and sb.value == 64 * per_chunk and wb.value > 0
    assert lib.shine_importance_sweep_sizes(3, rows3, 5, 4096, 0, C.byref(group), C.byref(sb), C.byref(wb)) == 0
    assert group.value == 5 and sb.value == 5 * per_chunk
    assert lib.shine_importance_sweep_sizes(3, rows3, 50, 4096, 3 * per_chunk + 7, C.byref(group), C.byref(sb), C.byref(wb)) == 0
    assert group.value == 3
    assert lib.shine_importance_sweep_sizes(3, rows3, 50, 4096, 1, C.byref(group), C.byref(sb), C.byref(wb)) == 0
    assert group.value == 1 and sb.value == per_chunk
    assert lib.shine_importance_sweep_sizes(5, rows3, 50, 4096, 0, C.byref(group), C.byref(sb), C.byref(wb)) == -1
    cfg = _lib.StepConfig()
    cfg.n_levels = 3
    assert lib.shine_train_step_workspace_bytes(C.byref(cfg), 4096) > 0
    info = (C.c_int64 * 8)()
    assert lib.shine_train_step_info(C.byref(cfg), 1 << 18, info) == 0
    assert info[0] > 0 and info[2] in (16, 32) and info[4] <= 160 * 1024  # workgroups, tile points, LDS per workgroup
    cfg.n_levels = 9
    assert lib.shine_train_step_info(C.byref(cfg), 1 << 18, info) == -1


def test_dropin_registers_the_reference_import_paths():
    """`import shine_mapping_amd.dropin` makes `from model.feature_octree import FeatureOctree` /
    `from model.decoder import Decoder` (shine_batch.py:13-14) resolve to this package, and nothing else."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import shine_mapping_amd.dropin as d\n"
        "from model.feature_octree import FeatureOctree\n"
        "from model.decoder import Decoder\n"
        "import shine_mapping_amd as s\n"
        "assert FeatureOctree is s.FeatureOctree and Decoder is s.Decoder\n"
        "d.uninstall()\n"
        "assert 'model.feature_octree' not in sys.modules\n"
        "print('ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_product_library_holds_the_fused_step_only_and_the_check_library_the_rest():
    """the product library ships ONE fused-step kernel generation.  The lane-per-point reference kernel
    (kernel_variant 1) lives in libshine_check.so, which tests / tools load explicitly; the product's dispatcher refuses that
    variant (and unplanned batches) instead of silently falling back.  Round 4 deleted the role-specialised experiment
    (a losing A/B of round 3) from the tree; round 5 gave kernel_variant 5 / 6 to the far / near build of the ONE fused step."""
    from shine_mapping_amd import _lib, build

    build.build(verbose=False)
    prod, chk = ctypes.CDLL(build.LIB), ctypes.CDLL(build.CHECK_LIB)
    assert hasattr(prod, "shine_train_step_v3") and not hasattr(prod, "shine_train_step_v0")
    assert hasattr(chk, "shine_train_step_v0") and hasattr(chk, "shine_train_step_v3")
    for gone in ("shine_train_step_v1", "shine_train_step_v2", "shine_train_step_v5"):
        assert not hasattr(prod, gone) and not hasattr(chk, gone)
    cfg = _lib.StepConfig()
    cfg.n_levels, cfg.max_level = 3, 12
    lib = _lib.lib()
    for variant, word in ((1, b"check library"), (7, b"unknown kernel_variant"), (0, b"plan"), (5, b"plan"), (6, b"plan")):
        cfg.kernel_variant = variant
        rc = lib.shine_train_step(None, ctypes.byref(cfg), None, None, None, None, None, None, 16, None, None, None, None,
                                  None, None, None, None, None, None, 0, None)
        assert rc == -1 and word in lib.shine_error_string(rc), (variant, lib.shine_error_string(rc))


def test_committed_counter_files_belong_to_the_kernels_of_this_build():
    """bench.py reports PMC-derived roofline fields only while the loaded library still contains the kernel the counters were
    measured on (tools/kernel_hash.py: sha256 of the kernel's machine code inside the .so).  The counter files committed under
    profiles/ for this round must name the fused kernels of THIS source tree — a kernel change without a fresh collection
    (tools/collect_profiles.sh) shows up here instead of silently dropping the measured fields from the bench line."""
    import json
    import sys

    from shine_mapping_amd import build

    build.build(verbose=False)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_hash

    seen = 0
    for workload, points, levels in (("maicity", 262144, 4), ("kitti", 1048576, 3), ("kitti-large", 1048576, 3)):
        import glob

        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s_%d_L%d.json" % (workload, points, levels))))
        assert found, (workload, points, levels)
        path = found[-1]  # the latest round's collection is the one bench.py reads
        rec = json.load(open(path))
        have = kernel_hash.step_kernel_sha256(workload, levels, build.LIB)
        assert have is not None and rec["kernel_code_sha256"] == have, (workload, rec["kernel_code_sha256"][:16], (have or "")[:16])
        assert rec["workload"] == workload and int(rec["points"]) == points and int(rec["levels"]) == levels
        assert rec["hbm_bytes_per_launch"] > 0 and 0 < rec["mfma_util"] < 1 and 0 < rec["fp32_datapath_util"] < 1
        seen += 1
    assert seen == 3


def test_device_constants_on_the_host():
    """_lib.device_constants (the optimiser's step state / learning rates, the sampler's stream id): a fill for equal values, a plain
    tensor on a CPU device (the pinned non-blocking copy is for CUDA devices only)."""
    import pytest
    import torch

    from shine_mapping_amd import _lib

    a = _lib.device_constants([0.01, 0.01, 0.01], torch.float32, "cpu")
    b = _lib.device_constants([3, 0, 3, 0], torch.int64, torch.device("cpu"))
    assert a.tolist() == pytest.approx([0.01] * 3) and a.dtype == torch.float32
    assert b.tolist() == [3, 0, 3, 0] and b.dtype == torch.int64
    assert _lib.device_constants([], torch.float32, "cpu").numel() == 0


def test_extension_build_failure_is_remembered_and_not_fatal(tmp_path, monkeypatch):
    """lib/_shine_ext.so (Tier A's C++ autograd nodes) is optional.  With a host compiler that fails, build()
    still returns the HIP library, warns once, leaves no stale extension behind and remembers the failure for this input (its own
    stamp: sources' digest + torch version + ABI flag) instead of recompiling everything at every call.  Runs on copies of the
    stamps in a scratch directory: the tree's own libraries are not touched."""
    import shutil
    import warnings

    import pytest

    from shine_mapping_amd import build

    build.build(verbose=False)  # (the tree is up to date)
    real_stamp = os.path.join(build.LIBDIR, "libshine_hip.stamp")
    monkeypatch.setattr(build, "LIBDIR", str(tmp_path))
    monkeypatch.setattr(build, "EXT_LIB", str(tmp_path / "_shine_ext.so"))
    shutil.copy(real_stamp, tmp_path / "libshine_hip.stamp")
    monkeypatch.setenv("CXX", "/bin/false")
    with pytest.warns(UserWarning, match="was not built"):
        assert build.build(verbose=False) == build.LIB
    assert not os.path.exists(build.EXT_LIB)
    assert open(tmp_path / "_shine_ext.stamp").read().endswith(":failed")
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # remembered: no second attempt, no second warning
        assert build.build(verbose=False) == build.LIB
    # another torch build / ABI flag is another input: the extension's stamp no longer matches
    assert build._ext_digest("x") != build._ext_digest("y")
