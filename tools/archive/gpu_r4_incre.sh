#!/bin/bash
# round 4, second pass (lab book block 9): the incremental frame — sweep / update tests, the ncd-incre bench line three times, the
# octree-update breakdown, one frame's device timeline (profiles/r04_timeline_ncd-incre_frame.txt) and the host profile of the loop
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04_incre; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "importance or incremental or octree or update or rebound" > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_part.log
for i in 1 2 3; do
  timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs > $O/bench_ncd_$i.json.log 2> $O/bench_ncd_$i.err
done
python - <<'PY'
import json
for u in (1, 2, 3):
    for l in open("gpurun_out/r04_incre/bench_ncd_%d.json.log" % u):
        if l.startswith("{"):
            r = json.loads(l); print("ncd", "%.1f fps" % r["frames_per_s"], r.get("iteration_graph"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items() if k != "note"}, {k: round(v, 3) for k, v in r["per_frame_host_issue_ms_median"].items()})
PY
timeout 300 python tools/update_breakdown.py > $O/update_breakdown.txt 2>&1; tail -3 $O/update_breakdown.txt
timeout 900 python -m cProfile -o /tmp/ncd.prof bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --steps 40 > /dev/null 2>&1
python - > $O/host_profile.txt 2>&1 <<'PY'
import pstats
p = pstats.Stats("/tmp/ncd.prof")
p.sort_stats("cumulative").print_stats("shine_mapping_amd|bench.py", 60)
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/tl_incre -o run -- python $R/bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs > $O/bench_trace_incre.log 2>&1
python $R/tools/frame_timeline.py /tmp/tl_incre > $O/frame_timeline.txt 2>&1; tail -25 $O/frame_timeline.txt
