#!/bin/bash
# round 5, call N: the single-launch plan of <= 4096-point batches — its test, the whole GPU suite, Tier A host cost and loop times
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_launch_plan or plan_batch" 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 240 python tools/tier_a_hostcost.py maicity > $O/tier_a_hostcost.log 2>&1; head -12 $O/tier_a_hostcost.log
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; tail -6 $O/tier_a_bench.log
