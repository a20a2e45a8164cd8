#!/bin/bash
# round 5, call Q: cal_regularization as a C++ node — GPU suite, config 4's Tier A host cost and frame rate
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 240 python tools/tier_a_hostcost.py incre > $O/tier_a_hostcost_incre.log 2>&1; grep -v amdgpu $O/tier_a_hostcost_incre.log | tail -13
SHINE_TIER_A_EXT_REG=1 TIER_A_SMALL=1 timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep "N=4096\|ncd-incre" $O/tier_a_bench.log
