#!/bin/bash
# round 3, call G: pipelined draw (bench + loop.GraphedIteration): correctness subset + A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03g; O=gpurun_out/r03g
timeout 600 python -m pytest tests -m gpu -x -q -k "graphed or unrolled or tier_a or deterministic" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
for wl in maicity kitti; do
  for f in "" "--no-pipeline"; do
    echo "== $wl $f"; timeout 300 python bench.py --workload $wl --no-extra-configs --no-cpu-baseline $f 2>$O/err_$wl.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline'].get('kernel_ms'), r['config'].get('launch'))"
  done
done
for f in "" "--no-pipeline"; do
  echo "== ncd-incre $f"; timeout 300 python bench.py --workload ncd-incre --no-cpu-baseline $f 2>>$O/err_incre.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['frames_per_s'], r['per_frame_ms_median'], r.get('like_for_like'))"
done
