#!/bin/bash
# one gpurun call: the bench lines of every workload (logs under gpurun_out/r02) + kernel stats / HBM PMC of kitti-large
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 600 python bench.py > $O/bench_maicity.json 2> $O/bench_maicity.err
timeout 600 python bench.py --workload kitti > $O/bench_kitti.json 2> $O/bench_kitti.err
timeout 600 python bench.py --workload ncd-incre > $O/bench_ncd.json 2> $O/bench_ncd.err
timeout 600 python bench.py --levels 3 --no-cpu-baseline > $O/bench_maicity_L3.json 2> $O/bench_maicity_L3.err
timeout 600 python bench.py --points 4096 --no-cpu-baseline > $O/bench_maicity_4096.json 2> $O/bench_maicity_4096.err
timeout 900 python bench.py --workload kitti-large --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_kitti_large.json 2> $O/bench_kitti_large.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --force-dist --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --force-dist --exchange touched --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dist1_touched.json 2> $O/bench_dist1_touched.err
(cd /tmp && export TMPDIR=/tmp && R=$OLDPWD && for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $C -d /tmp/pkl_$C -o run -- python $R/bench.py --workload kitti-large --no-cpu-baseline --steps 6 --warmup 2 --no-graph > $R/$O/pmc_kl_$C.log 2>&1; done; python $R/tools/pmc_to_json.py --kernel k_step_v --out $R/$O/pmc_kitti-large_1048576_L3.json --meta workload=kitti-large points=1048576 levels=3 -- /tmp/pkl_FETCH_SIZE /tmp/pkl_WRITE_SIZE > $R/$O/pmc_kitti_large.txt 2>&1)
tail -c 300 $O/*.err | tail -40
for f in $O/bench_*.json; do echo $f; head -c 700 $f; echo; done
