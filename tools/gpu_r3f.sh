#!/bin/bash
# gpurun call F of round 3: k_step_v3 with the stream dealt to the waves in interleaved chunks of 1 / 2 / 4 tiles
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
AB_VARIANTS=4 AB_PROF=1 AB_ONLY=maicity:4,kitti:3 timeout 1200 python tools/ab_build.py shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_c1.so tools/ab/lib_c2.so tools/ab/lib_c4.so > $O/ab_v3ch.txt 2>&1; grep -v amdgpu $O/ab_v3ch.txt | tail -14
