#!/bin/bash
# PMC passes of the kitti-large workload with the built map cached on the box's disk between the profiler's processes
cd "$GRAFT_REPO_ROOT"
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
( time timeout 1500 bash tools/collect_profiles.sh kitti-large 1048576 3 ) 2>&1 | tail -12
ls -la /tmp/shine_wl_cache; tail -3 gpurun_out/prof/pmc_kitti-large_1048576_L3.txt
