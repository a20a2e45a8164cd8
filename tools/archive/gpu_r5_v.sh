#!/bin/bash
# round 5, call V: rocprofv3 kernel stats + PMC passes for the two default workloads on SURVEY 8(d)'s full scan recipe
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05v; mkdir -p $O
timeout 900 bash tools/collect_profiles.sh maicity 262144 4 > $O/collect_maicity.log 2>&1; tail -2 $O/collect_maicity.log
timeout 1200 bash tools/collect_profiles.sh kitti 1048576 3 > $O/collect_kitti.log 2>&1; tail -2 $O/collect_kitti.log
cp gpurun_out/prof/*maicity_262144_L4* gpurun_out/prof/*kitti_1048576_L3* $O/ 2>/dev/null
head -12 $O/kernel_stats_maicity_262144_L4.txt; head -12 $O/kernel_stats_kitti_1048576_L3.txt
