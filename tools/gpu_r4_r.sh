#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04r; mkdir -p $O
timeout 300 python tools/tier_a_profile.py maicity 300 > $O/tier_a_profile_maicity.txt 2>&1
timeout 300 python tools/tier_a_profile.py kitti 300 > $O/tier_a_profile_kitti.txt 2>&1
grep -v amdgpu $O/tier_a_profile_maicity.txt | cut -c1-170 | head -120
