#!/bin/bash
# round 4, call k: fused appends + flat fetch in FeatureOctree.update, 10 iterations per graph
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "octree or update or importance or incremental or grow or pickle or checkpoint" > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_part.log
for u in 10 5; do
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --unroll $u > $O/bench_ncd_u$u.json.log 2> $O/bench_ncd_u$u.err
done
python - <<'PY'
import json
for u in (10, 5):
    for l in open("gpurun_out/r04k/bench_ncd_u%d.json.log" % u):
        if l.startswith("{"):
            r = json.loads(l); print("ncd unroll", u, "%.1f fps" % r["frames_per_s"], r.get("per_frame_total_ms"), r.get("iteration_graph"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items() if k != "note"}, {k: round(v, 3) for k, v in r["per_frame_host_issue_ms_median"].items()})
PY
timeout 300 python tools/update_breakdown.py > $O/update_breakdown.txt 2>&1; tail -3 $O/update_breakdown.txt
