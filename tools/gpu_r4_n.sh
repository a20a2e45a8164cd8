#!/bin/bash
# round 4, call n: diagnostics of the split step's second launch (no atomics / no walk / fewer workgroups)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04n; mkdir -p $O
AB_ONLY="maicity:4" timeout 900 python tools/ab_build.py tools/ab_run/lib_base.so tools/ab_run/lib_split.so tools/ab_run/lib_split_noatom.so tools/ab_run/lib_split_nowalk.so tools/ab_run/lib_split_w2.so > $O/ab_split_diag.txt 2>&1
cat $O/ab_split_diag.txt | grep -v amdgpu
