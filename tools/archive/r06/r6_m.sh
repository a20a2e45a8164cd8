#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06m; mkdir -p $O
PROFILE=1 timeout 300 python tools/tier_a_hostcost.py incre > $O/tier_a_hostcost_incre_profile.log 2>&1; grep -v amdgpu $O/tier_a_hostcost_incre_profile.log | head -75 | cut -c1-170
