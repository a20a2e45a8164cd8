#!/bin/bash
# round 6, call L: kernel trace of the two-launch step against the three-launch step (kitti, maicity)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r06l; mkdir -p $O
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
cd /tmp && export TMPDIR=/tmp
for w in kitti maicity; do
 for mode in rider norider; do
  extra=""; [ $mode = norider ] && extra="--no-draw-rider"
  rocprofv3 --kernel-trace --stats -d /tmp/p_${w}_$mode -o run -- python $R/bench.py --workload $w --no-extra-configs --no-cpu-baseline $extra --steps 200 --warmup 10 > $O/bench_${w}_$mode.log 2>&1
  python $R/tools/prof_summary.py /tmp/p_${w}_$mode 8 > $O/kernel_stats_${w}_$mode.txt 2>&1
  python $R/tools/timeline_gaps.py /tmp/p_${w}_$mode k_step_v3 20 > $O/timeline_${w}_$mode.txt 2>&1
  echo "== $w $mode"; head -14 $O/timeline_${w}_$mode.txt | cut -c1-150
 done
done
