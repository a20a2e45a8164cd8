#!/bin/bash
# gpurun call: the GPU suite and smoke() of the current tree -> gpurun_out/r03/pytest_gpu.log
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
