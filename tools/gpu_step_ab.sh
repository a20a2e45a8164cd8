#!/bin/bash
# whole-step A/B of two library builds on one box: bench.py (graph-replayed steps) alternately with each library in place
cd $GRAFT_REPO_ROOT 2>/dev/null || true
for rep in 1 2; do for v in "$@"; do cp tools/ab/lib_$v.so shine_mapping_amd/lib/libshine_hip.so; python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'step us %.2f' % (1e3*d['ms_per_step']), 'kernel us %.2f' % (1e3*d['roofline']['kernel_ms']))"; done; done
