#!/bin/bash
# gpurun call: the small-batch / incremental regime after the iteration hooks + MARK build
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "adam or graph or loop or regular or incre or importance or marks or tier_a" > $O/pytest_small.log 2>&1; echo "pytest rc=$?" >> $O/pytest_small.log; tail -6 $O/pytest_small.log
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline > $O/ncd_hooks.json 2> $O/ncd_hooks.err; python - <<PY
import json
d=json.loads([l for l in open("$O/ncd_hooks.json") if l.startswith("{")][-1])
print("ncd-incre: %.1f frames/s, %.1f us/iteration, split %s" % (d["frames_per_s"], d["us_per_iteration"], {k: round(v,2) for k,v in d["per_frame_ms_median"].items()}))
PY
timeout 900 python bench.py --no-extra-configs > $O/bench_maicity_hooks.json 2> $O/bench_maicity_hooks.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_maicity_hooks.json") if l.startswith("{")][-1])
print("maicity: step %.1f us, kernel %.1f us, like_for_like %s" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d.get("like_for_like",{}).get("gpu")))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["thread_calibration_ms"])
PY
