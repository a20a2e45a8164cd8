#!/bin/bash
# round 5, call F: Tier A once more (gradient views adopted by AccumulateGrad; backward on the calling thread), then the round's
# profiles: rocprofv3 kernel stats + PMC passes of the three headline kernels (tools/collect_profiles.sh)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05f; mkdir -p $O
TIER_A_SMALL=1 timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -4
timeout 600 python -m pytest tests -q -m gpu -k "tier_a or drop_in or fused_node or speculat or trajectory_tier or tier_a_incremental" > $O/pytest_tier_a.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_tier_a.log
for spec in "maicity 262144 4" "kitti 1048576 3" "kitti-large 1048576 3"; do
  timeout 900 bash tools/collect_profiles.sh $spec > $O/collect_$(echo $spec | tr ' ' '_').log 2>&1
done
mkdir -p $O/prof; cp gpurun_out/prof/*.txt gpurun_out/prof/*.json $O/prof/ 2>/dev/null
ls $O/prof | head -40
for f in $O/prof/kernel_stats_*.txt; do echo == $f; grep -E "k_step_v3|k_reduce|k_sample|Name" $f | head -6; done
