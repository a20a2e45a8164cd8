#!/bin/bash
# gpurun call: GPU suite, default bench line, profile collection of the default workload
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_maicity.json 2> $O/bench_maicity.err
cat $O/bench_maicity.json
timeout 900 bash tools/collect_profiles.sh maicity 262144 4 > $O/collect_maicity.log 2>&1
tail -5 $O/collect_maicity.log
