#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05r; mkdir -p $O
timeout 240 python tools/tier_a_hostcost.py eik > $O/tier_a_hostcost_eik.log 2>&1; grep -v amdgpu $O/tier_a_hostcost_eik.log | tail -16
