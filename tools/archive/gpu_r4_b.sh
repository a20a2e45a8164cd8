#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b; mkdir -p $O
timeout 600 python tools/incre_trajectory_debug.py > $O/incre_debug.txt 2>&1; tail -14 $O/incre_debug.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log
