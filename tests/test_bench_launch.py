"""CPU suite: `bench.py --gpus N` must MEAN N ranks.

The reference is single-GPU (utils/tools.py:26); the 1/2/4/8-GPU curve of BASELINE.json is measured by the driver with
`python bench.py --gpus N` or the explicit torchrun form.  Both must run the same thing: without a torchrun environment
the script re-launches itself under torch.distributed.run with N ranks, and it refuses to run with fewer devices than
ranks instead of silently benchmarking one GPU.  `--launch-check` stops after the rendezvous (gloo here, RCCL on a GPU
node), so the launcher is testable without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=env, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"], {"SHINE_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["world_size_reported"] == 2
    assert rec["config"]["parallelism"] == "dp2" and rec["backend"] == "gloo"
    assert "launching 2 ranks" in r.stderr


def test_gpus_1_stays_one_process():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["parallelism"] == "dp1"
    assert "launching" not in r.stderr


def test_too_few_devices_is_an_error_not_a_fallback():
    import torch

    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing to fall back" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no bench line labelled with fewer GPUs


def test_gpus_must_match_the_torchrun_world():
    r = _run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_explicit_torchrun_form_matches():
    """the driver's own command line: torch.distributed.run in front, --gpus N repeated after bench.py"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SHINE_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_printed_record_stays_small_and_strict():
    """round 5's record was ONE 22 kB line with four legs embedded and the driver could not parse it.  The
    formatter (bench.compact / compact_dp_rank / emit) on that very record — the canned full record of a real default run,
    profiles/r05_bench_default.json.log — must give lines of < 4096 bytes that strict JSON parsers accept, carrying the
    contract's fields, `roofline` and `cpu_baseline`; and hostile values (NaN, long strings) must not break the line."""
    import io
    from contextlib import redirect_stdout

    bench = _load_bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json.log")).read().strip().splitlines()[-1])
    legs = full.pop("configs")
    buf = io.StringIO()
    with redirect_stdout(buf):
        for name, leg in legs.items():
            line = bench.compact_dp_rank(leg) if name == "kitti-dp8-rank" else bench.compact(leg)
            bench.emit({"leg": name, "full_record": None, **line})
        bench.emit(bench.compact(full, "gpurun_out/bench_records/maicity.json"))
    lines = buf.getvalue().splitlines()
    assert len(lines) == 5 and sum(len(ln) for ln in lines) < 8192
    for ln in lines:
        assert len(ln) < bench.RECORD_LIMIT
        json.loads(ln, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # strict: no NaN / Infinity
    last = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in last, k
    assert last["value"] == float("%.5g" % full["value"]) and "workload" in last["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"} <= set(last["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(last["cpu_baseline"])
    assert [json.loads(ln).get("leg") for ln in lines[:-1]] == list(legs)
    # hostile values
    bad = dict(full)
    bad["config"] = dict(full["config"], workload="x" * 9000, launch="y" * 9000)
    bad["cpu_baseline"] = dict(full["cpu_baseline"], sample="z" * 9000, value=float("nan"))
    bad["roofline"] = dict(full["roofline"], frac=float("inf"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(bench.compact(bad))
    ln = buf.getvalue().strip()
    assert len(ln) < bench.RECORD_LIMIT and json.loads(ln)["roofline"]["frac"] is None
