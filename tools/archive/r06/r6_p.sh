#!/bin/bash
# round 6, call P: what the fused kernel's atomics / row gathers hold exclusively (measurement builds -DSHINE_V3_ABLATE)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06p; mkdir -p $O tools/ab
export SHINE_WORKLOAD_CACHE=/tmp/shine_wl_cache
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for v in 16; do python tools/mk_variant.py abl$v -DSHINE_V3_ABLATE=$v shine_step_v3.hip > /dev/null 2>&1 & done; wait
for w in kitti-large kitti maicity; do
  echo "== $w"
  timeout 600 python tools/far_ablate.py $w 2>&1 | grep -v amdgpu | tail -1
  for v in 16; do timeout 600 python tools/run_with_lib.py tools/ab/lib_abl$v.so tools/far_ablate.py $w 2>&1 | grep -v "amdgpu\|run_with_lib" | tail -1 | sed "s/kernel of the loaded library/SHINE_V3_ABLATE=$v            /"; done
done | tee $O/ablate_wait.txt
