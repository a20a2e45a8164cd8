#!/bin/bash
# round 4, call m: A/B of the split step (fused kernel without scatter + k_scatter_split) against the product
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04m; mkdir -p $O
AB_ONLY="maicity:4,maicity:3" timeout 900 python tools/ab_build.py tools/ab_run/lib_base.so tools/ab_run/lib_split.so tools/ab_run/lib_split_w6.so tools/ab_run/lib_split12.so > $O/ab_split.txt 2>&1
cat $O/ab_split.txt | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
AB_ONLY="maicity:4" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ab_split -o run -- python $R/tools/ab_build.py $R/tools/ab_run/lib_split.so > $O/ab_split_trace.log 2>&1
python $R/tools/prof_summary.py /tmp/ab_split 12 > $O/ab_split_kernel_stats.txt 2>&1; cat $O/ab_split_kernel_stats.txt
