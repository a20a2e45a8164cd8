#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/incre_check; O=gpurun_out/incre_check
timeout 1200 python -m pytest tests -m gpu -x -q -k "iteration_tail or graphed or unrolled or importance or incre or regulariser" > $O/pytest_subset.log 2>&1; tail -15 $O/pytest_subset.log
echo "== ncd-incre"; timeout 300 python bench.py --workload ncd-incre --no-cpu-baseline 2>>$O/err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['frames_per_s'], r['per_frame_ms_median'], r.get('like_for_like'))"
tail -5 $O/err.log
