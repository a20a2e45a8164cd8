#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err; tail -4 $O/bench_default_driver_form.err
for w in maicity kitti-large; do
  timeout 900 python bench.py --workload $w --no-extra-configs > $O/bench_$w.json.log 2> $O/bench_$w.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/*.json.log')):
    for l in open(f):
        if l.startswith('{'):
            r=json.loads(l); roof=r.get('roofline') or {}
            print(f.split('/')[-1], '%.4g'%r['value'], '%.4f ms'%r['ms_per_step'], 'kernel', roof.get('kernel_ms'), (r.get('like_for_like') or {}).get('gpu'))
PY
