#!/bin/bash
# round 5, call A: GPU suite on the packed-fp32 / far builds, then the kernel A/Bs (r04 library against this round's, near and far
# builds, the far build's knobs) on the cache-resident maps and on kitti-large, then the row-layout experiment
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 500 python tools/ab_build.py tools/ab/lib_pk.so@6 tools/ab/lib_r04.so > $O/ab_near.txt 2>&1; grep -v "^$" $O/ab_near.txt | grep -v amdgpu | tail -8
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@6,5 tools/ab/lib_r04.so tools/ab/lib_t0.so@5 tools/ab/lib_t1.so@5 tools/ab/lib_p1.so@5 tools/ab/lib_p2.so@5 > $O/ab_far.txt 2>&1; grep -v "^$" $O/ab_far.txt | grep -v amdgpu | tail -12
timeout 900 python tools/experiments/row_layout.py kitti_large 2800 300 > $O/row_layout.txt 2>&1; grep -v amdgpu $O/row_layout.txt | tail -12
