#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -30 $O/pytest_gpu.log
timeout 900 python tools/iter4096_bench.py maicity kitti-large > $O/iter4096.txt 2>&1; grep -v amdgpu $O/iter4096.txt | tail -8
