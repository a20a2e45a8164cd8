#!/bin/bash
# gpurun call: PMC passes (maicity, kitti), phase cycles of the default kernel, bench lines of the other workloads
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 600 bash tools/collect_profiles.sh maicity 262144 4 > $O/collect_maicity.log 2>&1
timeout 900 bash tools/collect_profiles.sh kitti 1048576 3 > $O/collect_kitti.log 2>&1
timeout 300 python tools/ablate.py > $O/ablate_maicity.txt 2>&1
timeout 600 python bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err
timeout 600 python bench.py --workload ncd-incre > $O/bench_ncd.json 2> $O/bench_ncd.err
tail -40 $O/ablate_maicity.txt
cat gpurun_out/prof/pmc_maicity_262144_L4.txt | head -60
head -c 1500 $O/bench_kitti.json; head -c 1500 $O/bench_ncd.json
