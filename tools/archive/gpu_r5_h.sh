#!/bin/bash
# round 5, call H: per-wave phase cycle counters (PROF instantiation) of the near and the far build on kitti-large and kitti
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05h; mkdir -p $O
AB_PROF=1 AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_prof.so@6,5 > $O/prof_far.txt 2>&1; grep -v "^$" $O/prof_far.txt | grep -v amdgpu | tail -6
AB_PROF=1 AB_ONLY=kitti:3 timeout 600 python tools/ab_build.py tools/ab/lib_prof.so@6,5 > $O/prof_near.txt 2>&1; grep -v "^$" $O/prof_near.txt | grep -v amdgpu | tail -6
