#!/bin/bash
# gpurun call B of round 3: k_step_v5 variants (wave priorities by role, interleaved chunk assignment, LDS-DMA rows) against
# k_step_v3, per-role counters of the best candidate, plain-VALU calibration modes of the micro-benchmark.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
date +%s > $O/t0
for V in pch4dma pch2dma; do
  timeout 600 python tools/run_with_lib.py tools/ab/lib_$V.so tools/v5_check.py > $O/v5_check_$V.txt 2>&1; echo "rc=$?" >> $O/v5_check_$V.txt; tail -3 $O/v5_check_$V.txt
done
timeout 900 python tools/run_with_lib.py tools/ab/lib_pch4dma.so -m pytest tests/test_gpu_scale_parity.py -m gpu -q -k "planned_ragged or regulariser_marks or weighted_bce or baseline_size" > $O/pytest_v5_pch4dma.log 2>&1; echo "pytest rc=$?" >> $O/pytest_v5_pch4dma.log; tail -4 $O/pytest_v5_pch4dma.log
AB_VARIANTS=4,5 AB_ONLY=maicity:4,kitti:3 timeout 1200 python tools/ab_build.py shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_mark.so tools/ab/lib_prio.so tools/ab/lib_ch4.so tools/ab/lib_ch8.so tools/ab/lib_pch4.so tools/ab/lib_pch4dma.so tools/ab/lib_pch2dma.so > $O/ab_v5b.txt 2>&1; grep -v "^  parity" $O/ab_v5b.txt | tail -6
timeout 600 python tools/v5_prof.py tools/ab/lib_pch4dmaprof.so > $O/v5_prof_b.txt 2>&1; tail -12 $O/v5_prof_b.txt
timeout 600 bash tools/gpu_calibrate.sh > $O/calibrate.log 2>&1; grep -A14 "cal_5\|cal_6\|cal_7" $O/ubench_calibration.txt | grep "==\|clock\|cycles per VALU\|ACTIVE_INST_VALU x\|MFMA busy /"
cat $O/ubench_plain.txt
echo "call wall $(( $(date +%s) - $(cat $O/t0) )) s"
