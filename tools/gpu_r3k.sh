#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03k; O=gpurun_out/r03k
for f in "" "--single-bucket"; do
  echo "== kitti-large $f"; timeout 600 python bench.py --workload kitti-large --no-extra-configs --no-cpu-baseline $f 2>$O/err_kl.log | tee -a $O/bench_kl.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline'].get('kernel_ms'), r['config'].get('launch'), r['config'].get('feature_table_bytes'))"
done
