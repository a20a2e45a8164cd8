#!/bin/bash
# round 6, call N: the data-parallel code path of bench.py over RCCL at world size 1 (torchrun form), every exchange; kitti too
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06n; mkdir -p $O
for ex in auto dense gather touched; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --exchange $ex --no-extra-configs --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_dist1_$ex.json.log 2> $O/bench_dist1_$ex.err
tail -1 $O/bench_dist1_$ex.json.log | cut -c1-300; grep -i "error\|Traceback" $O/bench_dist1_$ex.err | head -3
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --force-dist --workload kitti --points 524288 --exchange auto --micro-batches 2 --no-extra-configs --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_dist1_kitti.json.log 2> $O/bench_dist1_kitti.err
tail -1 $O/bench_dist1_kitti.json.log | cut -c1-600; grep -i "error\|Traceback" $O/bench_dist1_kitti.err | head -3
