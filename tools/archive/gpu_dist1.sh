#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dist1; O=gpurun_out/dist1
timeout 1200 python -m pytest tests -m gpu -x -q -k "own_rows or two_ranks or touched_row" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
for f in "" "--exchange gather --micro-batches 1"; do
  echo "== dist1 maicity $f"
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-dist --workload maicity --no-extra-configs --no-cpu-baseline $f 2>$O/err_dist1.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); c=r['config']; print(r['value'], r['ms_per_step'], c.get('launch'), '|', c.get('grad_exchange'), '|', c.get('grad_exchange_note'), c.get('grad_exchange_overflow'), c.get('grad_exchange_tuning_ms_per_step'))"
  grep -v "amdgpu.ids\|socket.cpp" $O/err_dist1.log | tail -5
done
echo "== plain"; timeout 300 python bench.py --no-extra-configs --no-cpu-baseline | cut -c1-300
