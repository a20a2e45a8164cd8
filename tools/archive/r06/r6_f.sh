#!/bin/bash
# round 6, call F: two-bucket step, C++ node lifetime tests, then kitti-large alone and the default run
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -2 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -x -k "two_gradient_buckets or cpp_nodes_hold" ) > $O/pytest_new.log 2>&1; grep -v "^$" $O/pytest_new.log | tail -25
timeout 900 python bench.py --workload kitti-large --no-extra-configs --no-cpu-baseline --full-record-dir $O/records > $O/bench_kitti-large.json.log 2> $O/bench_kitti-large.err; tail -2 $O/bench_kitti-large.err
python - <<PY
import json
r=json.loads(open("$O/bench_kitti-large.json.log").read().strip().splitlines()[-1])
print("kitti-large", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["config"]["launch"])
PY
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
