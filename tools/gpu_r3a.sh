#!/bin/bash
# gpurun call A of round 3: first contact of the role-specialised kernel (kernel_variant 5), A/B against k_step_v3 and the
# prepared v3 variants, per-role cycle counters, datapath calibration, the new default bench line, the full GPU suite.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
date +%s > $O/t0
timeout 600 python tools/v5_check.py > $O/v5_check.txt 2>&1; echo "v5_check rc=$?" >> $O/v5_check.txt; tail -12 $O/v5_check.txt
timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -k "planned_ragged or regulariser_marks or weighted_bce or baseline_size" > $O/pytest_v5.log 2>&1; echo "pytest rc=$?" >> $O/pytest_v5.log; tail -8 $O/pytest_v5.log
AB_VARIANTS=4,5 AB_ONLY=maicity:4,kitti:3 timeout 900 python tools/ab_build.py shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_predscat.so > $O/ab_v5.txt 2>&1; tail -8 $O/ab_v5.txt
timeout 600 python tools/v5_prof.py tools/ab/lib_v5prof.so > $O/v5_prof.txt 2>&1; tail -12 $O/v5_prof.txt
timeout 600 bash tools/gpu_calibrate.sh > $O/calibrate.log 2>&1; tail -40 $O/ubench_calibration.txt
T=$(date +%s); timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall $(( $(date +%s) - T )) s" | tee $O/bench_default.wall; cut -c1-600 $O/bench_default.json; tail -3 $O/bench_default.err
for V in base mark adamprep; do
  L=tools/ab/lib_$V.so; [ $V = base ] && L=shine_mapping_amd/lib/libshine_hip.so
  timeout 600 python tools/run_with_lib.py $L bench.py --workload ncd-incre --no-cpu-baseline > $O/ncd_$V.json 2> $O/ncd_$V.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/ncd_$V.json") if l.startswith("{")][-1])
    print("ncd-incre $V: %.1f frames/s, %.1f us/iteration, split %s" % (d["frames_per_s"], d["us_per_iteration"], {k: round(v,2) for k,v in d["per_frame_ms_median"].items()}))
except Exception as e:
    print("ncd-incre $V failed:", e)
PY
done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
echo "call wall $(( $(date +%s) - $(cat $O/t0) )) s"
