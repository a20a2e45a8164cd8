#!/bin/bash
# round 6, call G: in-kernel clear of the other gradient bucket — tests, kitti-large with / without two buckets
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -2 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -x -k "two_gradient_buckets or cpp_nodes_hold or far_build" ) > $O/pytest_new.log 2>&1; grep -v "^$" $O/pytest_new.log | tail -25
for tb in 1 0; do
SHINE_BENCH_TWO_BUCKETS=$tb timeout 900 python bench.py --workload kitti-large --no-extra-configs --no-cpu-baseline --full-record-dir $O/records$tb > $O/bench_kitti-large_$tb.json.log 2> $O/bench_kitti-large_$tb.err; tail -2 $O/bench_kitti-large_$tb.err
python - <<PY
import json
r=json.loads(open("$O/bench_kitti-large_$tb.json.log").read().strip().splitlines()[-1])
print("kitti-large two_buckets=$tb", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["config"]["launch"][-60:])
PY
done
