#!/bin/bash
# end-of-round gpurun call: GPU suite, smoke, every bench line of the round, Tier A iteration times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final; O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err; tail -4 $O/bench_default_driver_form.err
for w in maicity kitti ncd-incre kitti-large; do
  timeout 900 python bench.py --workload $w --no-extra-configs > $O/bench_$w.json.log 2> $O/bench_$w.err
done
timeout 600 python bench.py --workload ncd-incre --sync-frames --no-extra-configs --no-cpu-baseline > $O/bench_ncd-incre_sync_frames.json.log 2>/dev/null
timeout 600 python bench.py --workload maicity --levels 3 --no-extra-configs --no-cpu-baseline > $O/bench_maicity_L3.json.log 2>/dev/null
timeout 600 python bench.py --workload maicity --points 4096 --no-extra-configs --no-cpu-baseline > $O/bench_maicity_4096.json.log 2>/dev/null
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --no-extra-configs --no-cpu-baseline > $O/bench_dist1.json.log 2>/dev/null
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -8
timeout 300 python tools/update_breakdown.py > $O/update_breakdown.txt 2>&1; tail -3 $O/update_breakdown.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/*.json.log')):
    for l in open(f):
        if l.startswith('{'):
            r=json.loads(l); roof=r.get('roofline') or {}
            print(f.split('/')[-1], '%.4g'%r['value'], '%.4f ms'%r['ms_per_step'], 'kernel', roof.get('kernel_ms'), roof.get('bound'), roof.get('frac'), (roof.get('pmc') or {}).get('used'), 'fps', r.get('frames_per_s'), (r.get('like_for_like') or {}).get('gpu'))
PY
