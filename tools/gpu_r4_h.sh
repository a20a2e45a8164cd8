#!/bin/bash
# round 4, call h: the batched importance sweep (shine_sweep.hip) — its tests, the trajectory test, the ncd-incre bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "importance or incremental or rebound or regular" > $O/pytest_sweep.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sweep.log
for i in 1 2; do
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs > $O/bench_ncd_$i.json.log 2> $O/bench_ncd_$i.err
done
python - <<'PY'
import json
for i in (1, 2):
    for l in open("gpurun_out/r04h/bench_ncd_%d.json.log" % i):
        if l.startswith("{"):
            r = json.loads(l); print("ncd", "%.1f fps" % r["frames_per_s"], r.get("per_frame_total_ms"), r.get("iteration_graph"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items() if k != "note"}, {k: round(v, 3) for k, v in r["per_frame_host_issue_ms_median"].items()})
PY
tail -3 $O/bench_ncd_1.err
