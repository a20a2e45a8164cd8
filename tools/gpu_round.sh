#!/bin/bash
# one gpurun call: counters list + the bench lines of every workload (logs under gpurun_out/)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|GRBM_[A-Z_]*\|TCC_HIT[_a-z]*\|TCC_MISS[_a-z]*\|TCC_ATOMIC[_a-z]*\|TCC_REQ[_a-z]*\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u) > $O/counters.txt 2>&1
timeout 600 python bench.py > $O/bench_maicity.json 2> $O/bench_maicity.err
timeout 600 python bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err
timeout 600 python bench.py --workload ncd-incre > $O/bench_ncd.json 2> $O/bench_ncd.err
timeout 900 python bench.py --workload kitti-large --steps 50 --warmup 5 > $O/bench_kitti_large.json 2> $O/bench_kitti_large.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --force-dist --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --force-dist --exchange touched --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dist1_touched.json 2> $O/bench_dist1_touched.err
tail -c 600 $O/*.err
head -c 3000 $O/bench_*.json
