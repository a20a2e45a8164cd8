#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03q; O=gpurun_out/r03q
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for u in 1 3 4; do
echo "== ncd-incre unroll $u"; timeout 300 python bench.py --workload ncd-incre --no-cpu-baseline --unroll $u 2>>$O/err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['frames_per_s'], r['per_frame_ms_median'])"
done
