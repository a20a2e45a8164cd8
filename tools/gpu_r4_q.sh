#!/bin/bash
# round 4, call q: what would fewer feature-grad atomics buy the fused kernel? (a fraction of the run atomics skipped)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
AB_ONLY="maicity:4,kitti:3" timeout 900 python tools/ab_build.py tools/ab_run/lib_base.so tools/ab_run/lib_keep2.so tools/ab_run/lib_keep1.so tools/ab_run/lib_keep0.so > $O/ab_atomkeep.txt 2>&1
grep -v amdgpu $O/ab_atomkeep.txt
