#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rider; O=gpurun_out/rider
timeout 900 python -m pytest tests -m gpu -x -q -k "rides_on or surface or rank_slices or sorted_sampler or graphed" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
for wl in maicity kitti; do
  echo "== $wl"; timeout 300 python bench.py --workload $wl --no-extra-configs --no-cpu-baseline 2>$O/err_$wl.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], r['roofline'].get('kernel_ms'), r['roofline']['pmc'], r['config'].get('launch'), r['config'].get('preheat_ms'))"
done
echo "== driver form"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/err_default.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['ms_per_step'], {k:(v.get('value'),v.get('frames_per_s')) for k,v in r['configs'].items()})"
echo "== dist1"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-dist --workload maicity --no-extra-configs --no-cpu-baseline 2>$O/err_dist1.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); c=r['config']; print(r['value'], r['ms_per_step'], c.get('grad_exchange'), c.get('grad_exchange_tuning_ms_per_step'))"
