#!/bin/bash
# round 5, call C: the far build's corner lattice (closed node runs merged in LDS before they become atomics) against the near
# build and r04 on all three maps; the GPU suite; the host profile of a Tier A iteration
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR|allowance used" | tail -15
timeout 500 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so > $O/ab_near.txt 2>&1; grep -v "^$" $O/ab_near.txt | grep -v amdgpu | tail -8
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so > $O/ab_far.txt 2>&1; grep -v "^$" $O/ab_far.txt | grep -v amdgpu | tail -8
timeout 300 python tools/tier_a_profile.py maicity 300 > $O/tier_a_profile.txt 2>&1; grep -v amdgpu $O/tier_a_profile.txt | head -70
