#!/bin/bash
# round 5, call D: the GPU suite with Tier A's nodes in C++ (and once more with them in Python), the far build's lattice with the
# peek / close split against the near build and r04, Tier A iteration times (Python nodes vs C++ nodes, same process)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR|allowance used" | tail -15
SHINE_TIER_A_EXT=0 timeout 900 python -m pytest tests -q -m gpu -k "tier_a or drop_in or fused_node or speculat or bce or adam or trajectory_tier or decoder_forward" > $O/pytest_gpu_python_nodes.log 2>&1; echo "pytest (python nodes) rc=$?"; grep -v amdgpu $O/pytest_gpu_python_nodes.log | grep -E "passed|failed|FAILED|ERROR" | tail -8
timeout 500 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so > $O/ab_near.txt 2>&1; grep -v "^$" $O/ab_near.txt | grep -v amdgpu | tail -6
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so > $O/ab_far.txt 2>&1; grep -v "^$" $O/ab_far.txt | grep -v amdgpu | tail -4
TIER_A_SMALL=1 timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -4
