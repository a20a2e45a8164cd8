#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   kernel trace of the default bench command, PMC passes (each in its OWN run, --kernel-trace only), phase cycles.
# rocprofv3 databases stay in /tmp on the box; only text summaries land in gpurun_out/prof/ (copy them to profiles/).
set -u
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o run -- python $R/bench.py --steps 200 --warmup 10 > $OUT/bench_under_rocprof.log 2>&1
python $R/tools/prof_summary.py /tmp/p_trace 40 > $OUT/kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_trace_k -o run -- python $R/bench.py --workload kitti --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench_kitti_under_rocprof.log 2>&1
python $R/tools/prof_summary.py /tmp/p_trace_k 12 > $OUT/kernel_stats_kitti.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d /tmp/p_$C -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/pmc_$C.log 2>&1
  python $R/tools/pmc_summary.py /tmp/p_$C k_step_v1 > $OUT/pmc_$C.txt 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/p_sq -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py /tmp/p_sq k_step_v1 > $OUT/pmc_sq.txt 2>&1
cd $R
python tools/ablate.py > $OUT/ablate_and_phase_cycles.txt 2>&1
python bench.py > $OUT/bench_default.json.log 2>&1
python bench.py --workload kitti > $OUT/bench_kitti.json.log 2>&1
ls -la $OUT
