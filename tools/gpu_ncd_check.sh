#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02; mkdir -p $O
timeout 100 python -m pytest tests -m gpu -x -q -k "regulariser or sorted_sampler or incremental_loop or graphed_iteration or rank_slices or unrolled_graph" 2>&1 | tail -3
timeout 60 python bench.py --workload ncd-incre --no-cpu-baseline > $O/bench_ncd.json 2> $O/bench_ncd.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02/bench_ncd.json') if l.startswith('{')][-1])
print(d['per_frame_ms_median'], d['frames_per_s'], d['value'])
PY
