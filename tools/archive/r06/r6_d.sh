#!/bin/bash
# round 6, call D: the far regime's scatter stream without atomics (plain row RMW) against the atomics — tools/ubench/random_rows
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06d; mkdir -p $O tools/ab
hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ab/librandom_rows.so tools/ubench/random_rows.hip || exit 1
timeout 900 python tools/ubench/random_rows.py kitti_large 2800 300 1048576 > $O/ubench_random_rows_kitti_large.txt 2>&1; grep -v amdgpu $O/ubench_random_rows_kitti_large.txt | tail -12
timeout 600 python tools/ubench/random_rows.py kitti 120 450 1048576 > $O/ubench_random_rows_kitti.txt 2>&1; grep -v amdgpu $O/ubench_random_rows_kitti.txt | tail -12
