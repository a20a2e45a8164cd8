#!/bin/bash
# round 6, call V: the maicity profile again with the two-launch step (the bench's default since call K)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r06v; mkdir -p $O
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
( time timeout 1200 bash tools/collect_profiles.sh maicity 262144 4 ) > $O/collect_maicity.log 2>&1; tail -3 $O/collect_maicity.log
cp gpurun_out/prof/*maicity_262144_L4* $O/
head -9 $O/kernel_stats_maicity_262144_L4.txt; head -8 $O/timeline_maicity_262144_L4.txt
