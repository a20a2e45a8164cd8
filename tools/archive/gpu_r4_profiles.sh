#!/bin/bash
# round-4 profile collection: PMC + kernel stats + timelines of the three batch workloads (the built maps cached on the box's
# disk between the profiler's processes) -> gpurun_out/prof/  (copied to profiles/r04_*)
cd "$GRAFT_REPO_ROOT"
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
( time timeout 900 bash tools/collect_profiles.sh maicity 262144 4 ) 2>&1 | tail -4
( time timeout 900 bash tools/collect_profiles.sh kitti 1048576 3 ) 2>&1 | tail -4
( time timeout 1500 bash tools/collect_profiles.sh kitti-large 1048576 3 ) 2>&1 | tail -4
tail -3 gpurun_out/prof/pmc_*.txt
