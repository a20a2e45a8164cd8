#!/bin/bash
# round 4, second pass (lab book block 11): frames pipelined across the host / device boundary (asynchronous octree growth)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04_pipe; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "pipelined or incremental or importance or octree or update or graph" > $O/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_part.log
for mode in "" "--sync-frames" "" "--sync-frames"; do
  timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs $mode > $O/bench_ncd.json.log 2> $O/bench_ncd.err
  python - "$mode" <<'PY'
import json, sys
for l in open("gpurun_out/r04_pipe/bench_ncd.json.log"):
    if l.startswith("{"):
        r = json.loads(l); print("ncd", sys.argv[1] or "pipelined", "%.1f fps" % r["frames_per_s"], r.get("per_frame_total_ms"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items() if k != "note"}, {k: round(v, 3) for k, v in r["per_frame_host_issue_ms_median"].items()}, "loss %.6g" % r["final_loss"])
PY
  tail -2 $O/bench_ncd.err | grep -v amdgpu
done
