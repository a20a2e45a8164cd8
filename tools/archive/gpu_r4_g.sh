#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04g; mkdir -p $O
true
true
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --unroll 10 > $O/bench_ncd.json.log 2> $O/bench_ncd.err
python - <<'PY'
import json
for l in open("gpurun_out/r04g/bench_ncd.json.log"):
    if l.startswith("{"):
        r = json.loads(l); print("ncd", "%.1f fps" % r["frames_per_s"], r.get("per_frame_total_ms"), r.get("iteration_graph"), {k: round(v, 3) for k, v in r["per_frame_ms_median"].items()}, "ms/step %.3f" % r["ms_per_step"])
PY
