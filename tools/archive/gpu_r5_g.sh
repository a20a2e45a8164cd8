#!/bin/bash
# round 5, call G: the PMC passes of the far build on kitti-large (counters restricted to the fused step), then the round's final
# validation (tools/gpu_r5_final.sh)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05g; mkdir -p $O
SKIP_TRACE=1 timeout 1500 bash tools/collect_profiles.sh kitti-large 1048576 3 > $O/collect_kitti-large.log 2>&1; tail -3 $O/collect_kitti-large.log
cp gpurun_out/prof/pmc_kitti-large_1048576_L3.json gpurun_out/prof/pmc_kitti-large_1048576_L3.txt $O/ 2>/dev/null
bash tools/gpu_r5_final.sh
