#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04t; mkdir -p $O
timeout 900 python tools/iter4096_bench.py maicity > $O/iter4096.txt 2>&1; grep -v amdgpu $O/iter4096.txt | tail -4
timeout 900 python tools/iter4096_bench.py maicity >> $O/iter4096.txt 2>&1; grep -v amdgpu $O/iter4096.txt | tail -3
