#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
timeout 900 python -m pytest tests -m gpu -x -q -k "surface or graphed or unrolled or rank_slices or argument_checks" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
for wl in maicity kitti; do
  for f in "" "--graph-steps 1" "--graph-steps 8"; do
    echo "== $wl $f"; timeout 300 python bench.py --workload $wl --no-extra-configs --no-cpu-baseline $f 2>$O/err_$wl.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline'].get('kernel_ms'), r['config'].get('launch'))"
  done
done
