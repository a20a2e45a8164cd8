#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out/r03i; O=$R/gpurun_out/r03i
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for w in "maicity 262144 4" "kitti 1048576 3"; do
  set -- $w
  rocprofv3 --kernel-trace --stats -d /tmp/tl_$1 -o run -- python $R/bench.py --workload $1 --points $2 --levels $3 --no-cpu-baseline --no-extra-configs --steps 200 --warmup 10 > $O/bench_trace_$1.log 2>&1
  python $R/tools/timeline_gaps.py /tmp/tl_$1 k_step_v3 20 > $O/timeline_$1.txt 2>&1
  cat $O/timeline_$1.txt
done
