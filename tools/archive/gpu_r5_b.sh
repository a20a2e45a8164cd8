#!/bin/bash
# round 5, call B: the GPU suite (new: config-4-shaped trajectory with the freeze inside, Tier A frames, the far build on a
# > 256 MiB map, FusedAdam.state_dict), the far build = ids one tile ahead against the near build and r04 (+ nontemporal row
# gathers, + the 12-wave build), and the memory system's own time for the step's access stream (tools/ubench/random_rows.py)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR|allowance used" | tail -15
timeout 500 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so tools/ab/lib_w12.so@6 > $O/ab_near.txt 2>&1; grep -v "^$" $O/ab_near.txt | grep -v amdgpu | tail -8
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so tools/ab/lib_nt.so@5 > $O/ab_far.txt 2>&1; grep -v "^$" $O/ab_far.txt | grep -v amdgpu | tail -8
timeout 600 python tools/ubench/random_rows.py kitti_large 2800 300 > $O/random_rows_kitti_large.txt 2>&1; grep -v amdgpu $O/random_rows_kitti_large.txt | tail -6
timeout 300 python tools/ubench/random_rows.py kitti 120 450 > $O/random_rows_kitti.txt 2>&1; grep -v amdgpu $O/random_rows_kitti.txt | tail -6
