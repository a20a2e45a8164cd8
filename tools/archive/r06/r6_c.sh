#!/bin/bash
# round 6, call C: record-pool build — GPU suite, then the three batch workloads alone
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
for w in maicity kitti kitti-large; do
  timeout 900 python bench.py --workload $w --no-extra-configs --no-cpu-baseline --full-record-dir $O/records > $O/bench_$w.json.log 2> $O/bench_$w.err
  python - <<PY
import json
r=json.loads(open("$O/bench_$w.json.log").read().strip().splitlines()[-1])
print("$w", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])
PY
done
