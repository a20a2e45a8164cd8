#!/bin/bash
# round 4, call l: sweep fold per row / 64 chunks per launch
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "importance or incremental" > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_part.log
for i in 1 2 3; do
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs > $O/bench_ncd_$i.json.log 2> $O/bench_ncd_$i.err
done
python - <<'PY'
import json
for u in (1, 2, 3):
    for l in open("gpurun_out/r04l/bench_ncd_%d.json.log" % u):
        if l.startswith("{"):
            r = json.loads(l); print("ncd", "%.1f fps" % r["frames_per_s"], r.get("per_frame_total_ms"), r.get("iteration_graph"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items() if k != "note"}, {k: round(v, 3) for k, v in r["per_frame_host_issue_ms_median"].items()})
PY
