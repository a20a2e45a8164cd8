#!/bin/bash
# round 6, call T: small batches as one tile per workgroup — GPU suite, the geometry A/B, ncd-incre + the like-for-like iteration
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python tools/small_batch_geometry.py 2>&1 | grep -v amdgpu | tee $O/small_batch_geometry.txt
timeout 600 python bench.py --workload ncd-incre --no-extra-configs --full-record-dir $O/rec > $O/bench_ncd-incre.json.log 2>/dev/null; cut -c1-2600 $O/bench_ncd-incre.json.log | tail -1
timeout 600 python tools/iter4096_bench.py maicity kitti 2>&1 | grep -v amdgpu | tail -6
