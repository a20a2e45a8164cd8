#!/bin/bash
# Tier A (reference loop body on this package's autograd nodes) vs Tier B (fused step): iteration times + kernel stats,
# and the meshing query throughput
cd $GRAFT_REPO_ROOT 2>/dev/null || true
R=$PWD
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -m gpu -x -q -k "tier_a or interp or mlp or autograd or mesh or query or forward" 2>&1 | tail -3
true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_tiera -o run -- python $R/tools/tier_a_bench.py > $R/gpurun_out/r02/tier_a_bench.log 2>&1
grep -E "tier A" $R/gpurun_out/r02/tier_a_bench.log
python $R/tools/prof_summary.py /tmp/p_tiera 8 > $R/gpurun_out/r02/kernel_stats_tier_a.txt 2>&1
cut -c1-150 $R/gpurun_out/r02/kernel_stats_tier_a.txt
