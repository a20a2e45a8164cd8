#!/bin/bash
# gpurun call D of round 3: k_step_v5 with the early-issue gather wave (rows of tile j+1 requested right after tile j's were
# consumed; corners 4-7 by LDS-DMA), trash-row sums in the decoder waves, and the prefix-sum scatter (also in k_step_v3).
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
for V in n0 n3; do
  timeout 600 python tools/run_with_lib.py tools/ab/lib_$V.so tools/v5_check.py > $O/v5_check_$V.txt 2>&1; echo "rc=$?" >> $O/v5_check_$V.txt; tail -11 $O/v5_check_$V.txt
done
timeout 900 python tools/run_with_lib.py tools/ab/lib_n2.so -m pytest tests/test_gpu_scale_parity.py -m gpu -q -k "planned_ragged or regulariser_marks or weighted_bce or baseline_size" > $O/pytest_v5_n2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_v5_n2.log; tail -4 $O/pytest_v5_n2.log
SHINE_KERNEL_V3_ONLY=1 timeout 900 python tools/run_with_lib.py tools/ab/lib_v3prefix.so -m pytest tests/test_gpu_scale_parity.py tests/test_gpu_parity.py -m gpu -q -k "planned_ragged or baseline_size or golden or regulariser_marks" > $O/pytest_v3prefix.log 2>&1; echo "pytest rc=$?" >> $O/pytest_v3prefix.log; tail -4 $O/pytest_v3prefix.log
AB_VARIANTS=4,5 AB_ONLY=maicity:4,kitti:3 timeout 1200 python tools/ab_build.py shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_v3prefix.so tools/ab/lib_n0.so tools/ab/lib_n1.so tools/ab/lib_n2.so tools/ab/lib_n3.so tools/ab/lib_n4.so tools/ab/lib_n5.so tools/ab/lib_n6.so > $O/ab_v5d.txt 2>&1; grep -v "^  parity" $O/ab_v5d.txt | tail -6; grep "parity" $O/ab_v5d.txt | head -20
timeout 600 python tools/v5_prof.py tools/ab/lib_n2prof.so > $O/v5_prof_d.txt 2>&1; tail -12 $O/v5_prof_d.txt
