#!/bin/bash
# round 5, call I: the far build's tile share of the older waves (140/256 was tuned on the cache-resident eikonal map) and a dump
# of its per-wave phase counters
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05i; mkdir -p $O
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@5 tools/ab/lib_s128.so@5 tools/ab/lib_s116.so@5 tools/ab/lib_s152.so@5 > $O/ab_share.txt 2>&1; grep -v "^$" $O/ab_share.txt | grep -v amdgpu | tail -5
AB_PROF=1 AB_PROF_DUMP=$O AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_prof.so@5 > $O/prof_dump.txt 2>&1; ls -la $O
