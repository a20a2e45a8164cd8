#!/bin/bash
# gpurun call: rocprofv3 kernel stats + PMC passes for the named workloads ("maicity 262144 4" "kitti 1048576 3" ...)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out/r02
for W in "$@"; do
  set -- $W
  timeout 1200 bash tools/collect_profiles.sh $1 $2 $3 > gpurun_out/r02/collect_$1.log 2>&1
  head -c 1200 gpurun_out/prof/pmc_$1_$2_L$3.txt | head -5
done
ls gpurun_out/prof | head -40
