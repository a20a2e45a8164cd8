#!/bin/bash
# round 5, call P: FusedAdam's steady-state fast path — GPU suite, Tier A host cost, loop times
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05p; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 240 python tools/tier_a_hostcost.py maicity > $O/tier_a_hostcost.log 2>&1; head -12 $O/tier_a_hostcost.log | grep -v amdgpu
timeout 400 python tools/plan_small_ab.py maicity > $O/plan_small_ab.log 2>&1; grep "tier" $O/plan_small_ab.log
