#!/bin/bash
# round 6, call Q: the pipelined scatter — parity (golden + scale tests), then the three kernels alone against -DSHINE_V3_PIPE=0
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06q; mkdir -p $O tools/ab
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
python tools/mk_variant.py nopipe -DSHINE_V3_PIPE=0 shine_step_v3.hip shine_sweep.hip > /dev/null 2>&1 &
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale_parity.py -m gpu -q -x ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
wait
for w in maicity kitti kitti-large; do
  echo "== $w"
  timeout 600 python tools/far_ablate.py $w 2>&1 | grep -v amdgpu | tail -1 | sed "s/kernel of the loaded library/pipelined scatter (product)  /"
  timeout 600 python tools/run_with_lib.py tools/ab/lib_nopipe.so tools/far_ablate.py $w 2>&1 | grep -v "amdgpu\|run_with_lib" | tail -1 | sed "s/kernel of the loaded library/-DSHINE_V3_PIPE=0 (round 5's order)/"
done | tee $O/pipe_ab.txt
