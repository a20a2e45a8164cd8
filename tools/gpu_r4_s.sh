#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04s; mkdir -p $O
for i in 1 2; do
TIER_A_SMALL=1 timeout 300 python tools/tier_a_bench.py 2>&1 | grep -v amdgpu | sed "s/^/mt  /" | tee -a $O/tier_a_threads.txt
TIER_A_SMALL=1 TIER_A_SINGLE_THREAD_AUTOGRAD=1 timeout 300 python tools/tier_a_bench.py 2>&1 | grep -v amdgpu | sed "s/^/st  /" | tee -a $O/tier_a_threads.txt
done
