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
