#!/bin/bash
# round 5, call S: in-process A/B of gather variants of the fused step (tools/abx/*.so)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05s; mkdir -p $O
AB_ONLY=maicity:4,kitti:3 timeout 500 python tools/ab_build.py $AB_LIBS > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | tail -30
