#!/bin/bash
# round 5, call E: GPU suite (Tier A on the C++ nodes), smoke, the default bench line in the driver's form (now with the
# kitti-large and kitti-dp8-rank legs and Tier A's ncd-incre line), Tier A iteration / frame times
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR|allowance used" | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err; tail -4 $O/bench_default_driver_form.err | grep -v amdgpu
python - <<'PY'
import json
for l in open("gpurun_out/r05e/bench_default_driver_form.json.log"):
    if l.startswith("{"):
        r = json.loads(l); roof = r.get("roofline") or {}
        print("default: %.4g %s, %.4f ms/step, kernel %.4f ms, bound %s frac %.3f, pmc used %s" % (r["value"], r["unit"], r["ms_per_step"], roof.get("kernel_ms"), roof.get("bound"), roof.get("frac"), (roof.get("pmc") or {}).get("used")))
        for k, v in (r.get("configs") or {}).items():
            if "error" in v: print("  ", k, "ERROR", v["error"]); continue
            rf = v.get("roofline") or {}
            print("  ", k, v.get("value"), v.get("ms_per_step"), "kernel", rf.get("kernel_ms"), rf.get("bound"), rf.get("frac"), "fps", v.get("frames_per_s"), "tierA", (v.get("tier_a") or {}).get("frames_per_s"), (v.get("tier_a") or {}).get("error"), v.get("ms_per_step_measured"), (v.get("scale_model") or {}).get("by_exchange"))
PY
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -7
