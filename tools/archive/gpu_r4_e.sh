#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -8
timeout 300 python tools/forward_bench.py > $O/forward_bench.log 2>&1; tail -6 $O/forward_bench.log
