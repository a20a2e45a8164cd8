#!/bin/bash
# round 6, call E: distinct corner rows per window of the ordered stream on the 324 MB map (what ANY merge inside a wave's /
# a workgroup's range could reach) — the floor of the far regime's row updates
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06e; mkdir -p $O
timeout 900 python tools/atomics_count.py kitti_large 1048576 3 2800 300 > $O/atomics_count_kitti_large.txt 2>&1; grep -v amdgpu $O/atomics_count_kitti_large.txt | tail
