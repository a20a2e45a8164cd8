#!/bin/bash
# round 6, call H: rocprofv3 kernel stats + PMC passes (tools/collect_profiles.sh) for the three batch workloads on the
# record-pool build; the built maps are cached on the box's disk between the profiler's processes
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r06h; mkdir -p $O
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 1200 bash tools/collect_profiles.sh maicity 262144 4 ) > $O/collect_maicity.log 2>&1; tail -3 $O/collect_maicity.log
( time timeout 1800 bash tools/collect_profiles.sh kitti 1048576 3 ) > $O/collect_kitti.log 2>&1; tail -3 $O/collect_kitti.log
( time timeout 2400 bash tools/collect_profiles.sh kitti-large 1048576 3 ) > $O/collect_kitti-large.log 2>&1; tail -3 $O/collect_kitti-large.log
cp gpurun_out/prof/* $O/ 2>/dev/null
for t in maicity_262144_L4 kitti_1048576_L3 kitti-large_1048576_L3; do head -8 $O/kernel_stats_$t.txt; tail -4 $O/pmc_$t.txt; done
