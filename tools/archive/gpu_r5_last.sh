#!/bin/bash
# round 5, last call: what the driver runs at round end — GPU suite, smoke, the default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05last; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json.log 2> $O/bench_default.err; tail -4 $O/bench_default.err | grep real
python - <<'PY'
import json
for l in open("gpurun_out/r05last/bench_default.json.log"):
    if l.startswith("{"):
        r = json.loads(l); rf = r["roofline"]
        print("value %.4g %s, %.4f ms/step, kernel %.4f ms, roofline %s frac %.3f (pmc used %s), algorithmic %.3f, cpu_baseline %.4g %s" % (
            r["value"], r["unit"], r["ms_per_step"], rf["kernel_ms"], rf["bound"], rf["frac"], rf["pmc"]["used"], rf["algorithmic"]["frac_of_hbm_peak"], r["cpu_baseline"]["value"], r["cpu_baseline"]["unit"]))
        for k, v in r["configs"].items():
            print("  ", k, "ERROR " + v["error"] if "error" in v else (v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("kernel_ms"), (v.get("roofline") or {}).get("frac"), ((v.get("roofline") or {}).get("pmc") or {}).get("used"), v.get("frames_per_s"), (v.get("tier_a") or {}).get("frames_per_s")))
PY
