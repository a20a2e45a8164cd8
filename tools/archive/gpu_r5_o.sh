#!/bin/bash
# round 5, call O: batches of <= 16384 points planned in one launch, not reordered — GPU suite, same-box A/B, Tier A times
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05o; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 400 python tools/plan_small_ab.py > $O/plan_small_ab.log 2>&1; grep -v amdgpu.ids $O/plan_small_ab.log | tail -14
timeout 240 python tools/tier_a_hostcost.py maicity > $O/tier_a_hostcost.log 2>&1; head -12 $O/tier_a_hostcost.log | grep -v amdgpu
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep "N=4096\|ncd-incre" $O/tier_a_bench.log
