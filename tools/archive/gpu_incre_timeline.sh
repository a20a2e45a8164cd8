#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out/incre_tl; O=$R/gpurun_out/incre_tl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tl_incre -o run -- python $R/bench.py --workload ncd-incre --no-cpu-baseline > $O/bench_trace_incre.log 2>&1
python $R/tools/timeline_gaps.py /tmp/tl_incre k_step_v3 100 > $O/timeline_incre.txt 2>&1
python $R/tools/prof_summary.py /tmp/tl_incre 45 > $O/kernel_stats_incre.txt 2>&1
cat $O/timeline_incre.txt; tail -2 $O/bench_trace_incre.log | cut -c1-1500
