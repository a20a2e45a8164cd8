#!/bin/bash
# gpurun call: the GPU suite, smoke, the bench lines of BASELINE configs 2 / 3 / 4 and the profile passes of the final build
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/gpu_profiles.sh "maicity 262144 4" "kitti 1048576 3" > $O/profiles_final.log 2>&1
mkdir -p profiles_new; cp gpurun_out/prof/pmc_maicity_262144_L4.json profiles/r02_pmc_maicity_262144_L4.json; cp gpurun_out/prof/pmc_kitti_1048576_L3.json profiles/r02_pmc_kitti_1048576_L3.json
timeout 600 python bench.py > $O/bench_maicity.json 2> $O/bench_maicity.err
timeout 600 python bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err
timeout 600 python bench.py --workload ncd-incre > $O/bench_ncd.json 2> $O/bench_ncd.err
for f in maicity kitti ncd; do grep "^{" $O/bench_$f.json | cut -c1-260; done
