#!/bin/bash
# kernel trace (stats + per-step launch timeline) of the bench command, no counters: refreshes profiles/r0N_rocprofv3_kernel_stats_*
# and r0N_timeline_* after changes around the fused kernel
cd "$GRAFT_REPO_ROOT"; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in "maicity 262144 4" "kitti 1048576 3"; do
  set -- $w; TAG=${1}_${2}_L${3}
  rocprofv3 --kernel-trace --stats -d /tmp/p_trace_$TAG -o run -- python $R/bench.py --workload $1 --points $2 --levels $3 --no-cpu-baseline --no-extra-configs --steps 200 --warmup 10 > $OUT/bench_under_rocprof_$TAG.log 2>&1
  python $R/tools/prof_summary.py /tmp/p_trace_$TAG 30 > $OUT/kernel_stats_$TAG.txt 2>&1
  python $R/tools/timeline_gaps.py /tmp/p_trace_$TAG k_step_v3 20 > $OUT/timeline_$TAG.txt 2>&1
  head -12 $OUT/timeline_$TAG.txt
done
