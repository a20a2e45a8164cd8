#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
for U in 2 5 10 25; do
  timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --unroll $U > $O/bench_ncd_u$U.json.log 2> $O/bench_ncd_u$U.err
  python - <<PY
import json
for l in open("$O/bench_ncd_u$U.json.log"):
    if l.startswith("{"):
        r = json.loads(l); print("unroll $U", "%.1f fps" % r["frames_per_s"], {k: round(v, 3) for k, v in r["per_frame_ms_median"].items()}, "us/iter %.1f" % r["us_per_iteration"])
PY
done
