#!/bin/bash
# gpurun call (or part of one): PMC passes over tools/ubench/mfma_valu_overlap, one mode per process, to calibrate the
# fp32-datapath formula of tools/pmc_to_json.py / bench.py: on a kernel that issues ONLY MFMAs
# (modes 0, 3) or ONLY v_fma_f32 (modes 1, 4) the formula must return the known occupancy.
#   -> gpurun_out/r03/ubench_calibration.txt  (copy to profiles/r03_ubench_calibration.txt)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tools/ubench/bin/mfma_valu_overlap
$B > $O/ubench_plain.txt 2>&1
for M in 0 1 3 4 5 6 7; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/cal_$M -o run -- $B $M > $O/ubench_pmc_mode$M.log 2>&1
done
python $R/tools/calibrate_datapath.py /tmp/cal_0 /tmp/cal_1 /tmp/cal_3 /tmp/cal_4 /tmp/cal_5 /tmp/cal_6 /tmp/cal_7 > $O/ubench_calibration.txt 2>&1
cat $O/ubench_plain.txt $O/ubench_calibration.txt
