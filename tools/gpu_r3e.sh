#!/bin/bash
# gpurun call E of round 3: k_step_v5 with the chunks of a workgroup CLAIMED by its gather waves (SHINE_V5_DYN) against the
# static interleave and k_step_v3.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
for V in d1 d3; do
  timeout 600 python tools/run_with_lib.py tools/ab/lib_$V.so tools/v5_check.py > $O/v5_check_$V.txt 2>&1; echo "rc=$?" >> $O/v5_check_$V.txt; tail -3 $O/v5_check_$V.txt
done
timeout 900 python tools/run_with_lib.py tools/ab/lib_d1.so -m pytest tests/test_gpu_scale_parity.py -m gpu -q -k "planned_ragged or regulariser_marks or weighted_bce or baseline_size" > $O/pytest_v5_d1.log 2>&1; echo "pytest rc=$?" >> $O/pytest_v5_d1.log; tail -4 $O/pytest_v5_d1.log
AB_VARIANTS=4,5 AB_ONLY=maicity:4,kitti:3,maicity:3 timeout 1200 python tools/ab_build.py shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_d0.so tools/ab/lib_d1.so tools/ab/lib_d2.so tools/ab/lib_d3.so tools/ab/lib_d4.so > $O/ab_v5e.txt 2>&1; grep -v "^  parity" $O/ab_v5e.txt | tail -6
timeout 600 python tools/v5_prof.py tools/ab/lib_d1prof.so > $O/v5_prof_e.txt 2>&1; tail -12 $O/v5_prof_e.txt
