#!/bin/bash
# round 4, second pass (lab book block 10): the scatter as its own launch, and what the feature-grad atomics cost.  BEFORE the call:
#   git apply tools/experiments/r04_split_scatter.patch
#   mkdir -p tools/ab_run && cp shine_mapping_amd/lib/libshine_check.so tools/ab_run/lib_base.so      (built from the clean tree)
#   for v in "split -DSHINE_V3_SPLIT=1" "split_noatom -DSHINE_V3_SPLIT=1 -DSHINE_V3_SPLIT_DIAG=1" "split_nowalk -DSHINE_V3_SPLIT=1 -DSHINE_V3_SPLIT_DIAG=2"; do
#     AB_DIR=tools/ab_run python tools/mk_variant.py $v all; done
#   for k in 2 1 0; do AB_DIR=tools/ab_run python tools/mk_variant.py keep$k -DSHINE_V3_ATOMKEEP=$k; done
#   git apply -R tools/experiments/r04_split_scatter.patch          (tools/ab_run/ travels with gpurun; delete it afterwards)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04_block10; mkdir -p $O
A=tools/ab_run
AB_ONLY="maicity:4,maicity:3" timeout 900 python tools/ab_build.py $A/lib_base.so $A/lib_split.so $A/lib_split_noatom.so $A/lib_split_nowalk.so > $O/ab_split.txt 2>&1
AB_ONLY="maicity:4,kitti:3" timeout 900 python tools/ab_build.py $A/lib_base.so $A/lib_keep2.so $A/lib_keep1.so $A/lib_keep0.so > $O/ab_atomkeep.txt 2>&1
grep -v amdgpu $O/ab_split.txt $O/ab_atomkeep.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics_rows tools/ubench/atomics_rows.hip 2>/dev/null
timeout 120 /tmp/atomics_rows > $O/ubench_atomics_rows.txt 2>&1; cat $O/ubench_atomics_rows.txt
timeout 300 python tools/atomics_count.py maicity 262144 4 > $O/atomics_count.txt 2>&1
timeout 300 python tools/atomics_count.py kitti 1048576 3 >> $O/atomics_count.txt 2>&1; grep -v amdgpu $O/atomics_count.txt
cd /tmp && export TMPDIR=/tmp
AB_ONLY="maicity:4" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ab_split -o run -- python $R/tools/ab_build.py $R/$A/lib_split.so > $O/ab_split_trace.log 2>&1
python $R/tools/prof_summary.py /tmp/ab_split 6 > $O/ab_split_kernel_stats.txt 2>&1; cat $O/ab_split_kernel_stats.txt
