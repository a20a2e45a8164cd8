#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 bash tools/collect_profiles.sh maicity 262144 4
timeout 900 bash tools/collect_profiles.sh kitti 1048576 3
cat gpurun_out/prof/timeline_*.txt
