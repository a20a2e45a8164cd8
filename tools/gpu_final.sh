#!/bin/bash
# gpurun call: the GPU suite, smoke, the default bench line, Tier A vs Tier B iteration times, the meshing query
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; cat $O/tier_a_bench.log | tail -4
timeout 300 python tools/forward_bench.py 2>&1 | tail -4 > $O/forward_bench.log; cat $O/forward_bench.log
timeout 600 python bench.py > $O/bench_maicity.json 2> $O/bench_maicity.err
timeout 600 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline > $O/bench_ncd.json 2> $O/bench_ncd.err
for f in maicity kitti ncd; do grep "^{" $O/bench_$f.json | cut -c1-200; done
