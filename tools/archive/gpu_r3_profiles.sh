#!/bin/bash
# round-3 profile collection: PMC + kernel stats + timelines for the three batch workloads, then the bench logs
cd "$GRAFT_REPO_ROOT"; R=$PWD
timeout 900 bash tools/collect_profiles.sh maicity 262144 4 > /dev/null 2>&1
timeout 900 bash tools/collect_profiles.sh kitti 1048576 3 > /dev/null 2>&1
timeout 1200 bash tools/collect_profiles.sh kitti-large 1048576 3 > /dev/null 2>&1
mkdir -p gpurun_out/r03_bench; O=gpurun_out/r03_bench
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err
for w in maicity kitti kitti-large ncd-incre; do
  timeout 900 python bench.py --workload $w --no-extra-configs > $O/bench_$w.json.log 2> $O/bench_$w.err
done
timeout 600 python bench.py --workload maicity --levels 3 --no-extra-configs --no-cpu-baseline > $O/bench_maicity_L3.json.log 2>/dev/null
timeout 600 python bench.py --workload maicity --points 4096 --no-extra-configs --no-cpu-baseline > $O/bench_maicity_4096.json.log 2>/dev/null
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --no-extra-configs --no-cpu-baseline > $O/bench_dist1.json.log 2>/dev/null
tail -3 $O/bench_default_driver_form.err; ls -la $O gpurun_out/prof | head -60
