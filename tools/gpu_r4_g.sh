#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python tools/update_breakdown.py > $O/update_breakdown.txt 2>&1; tail -6 $O/update_breakdown.txt
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --unroll 10 > $O/bench_ncd.json.log 2> $O/bench_ncd.err
python - <<'PY'
import json
for l in open("gpurun_out/r04g/bench_ncd.json.log"):
    if l.startswith("{"):
        r = json.loads(l); print("ncd", "%.1f fps" % r["frames_per_s"], {k: round(v, 3) for k, v in r["per_frame_ms_median"].items()}, "ms/step %.3f" % r["ms_per_step"])
PY
