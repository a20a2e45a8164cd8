#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O
timeout 300 python tools/atomics_count.py maicity 262144 4 > $O/atomics_count_maicity.txt 2>&1; grep -v amdgpu $O/atomics_count_maicity.txt
timeout 300 python tools/atomics_count.py kitti 1048576 3 > $O/atomics_count_kitti.txt 2>&1; grep -v amdgpu $O/atomics_count_kitti.txt
