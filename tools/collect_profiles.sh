#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <workload> <points> <levels> [extra bench args]
#   kernel trace of the bench command, PMC passes (each in its OWN run, --kernel-trace only; gpurun refuses --pmc combined
#   with sys/hip/hsa traces).  rocprofv3 databases stay in /tmp on the box; the text/JSON summaries land in
#   gpurun_out/prof/ (copy them to profiles/, named per round).
set -u
W=${1:-maicity}; P=${2:-262144}; L=${3:-4}; shift 3 || true
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
TAG=${W}_${P}_L${L}
BENCH="python $R/bench.py --workload $W --points $P --levels $L --no-cpu-baseline --no-extra-configs $*"
cd /tmp && export TMPDIR=/tmp
# counters for the fused step only: on the 288 M-sample pool of kitti-large every kernel of the pool plan was otherwise
# serialised under the counters too (one SQ pass took 11 minutes in round 5's first collection)
KF="--kernel-include-regex k_step_v3"
if [ -z "${SKIP_TRACE:-}" ]; then
rocprofv3 --kernel-trace --stats -d /tmp/p_trace_$TAG -o run -- $BENCH --steps 200 --warmup 10 > $OUT/bench_under_rocprof_$TAG.log 2>&1
python $R/tools/prof_summary.py /tmp/p_trace_$TAG 30 > $OUT/kernel_stats_$TAG.txt 2>&1
python $R/tools/timeline_gaps.py /tmp/p_trace_$TAG k_step_v3 20 > $OUT/timeline_$TAG.txt 2>&1
fi
DIRS=""
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace $KF --pmc $C -d /tmp/p_${C}_$TAG -o run -- $BENCH --steps 6 --warmup 2 --no-graph > $OUT/pmc_${C}_$TAG.log 2>&1
  DIRS="$DIRS /tmp/p_${C}_$TAG"
done
rocprofv3 --kernel-trace $KF --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/p_sq_$TAG -o run -- $BENCH --steps 6 --warmup 2 --no-graph > $OUT/pmc_sq_$TAG.log 2>&1
rocprofv3 --kernel-trace $KF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma_$TAG -o run -- $BENCH --steps 6 --warmup 2 --no-graph > $OUT/pmc_mfma_$TAG.log 2>&1
rocprofv3 --kernel-trace $KF --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum -d /tmp/p_tcc_$TAG -o run -- $BENCH --steps 6 --warmup 2 --no-graph > $OUT/pmc_tcc_$TAG.log 2>&1
rocprofv3 --kernel-trace $KF --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d /tmp/p_tex_$TAG -o run -- $BENCH --steps 6 --warmup 2 --no-graph > $OUT/pmc_tex_$TAG.log 2>&1
python $R/tools/pmc_to_json.py --kernel k_step_v --out $OUT/pmc_$TAG.json --meta workload=$W points=$P levels=$L \
  --command "rocprofv3 --kernel-trace --pmc <group> -- $BENCH --steps 6 --warmup 2 --no-graph" \
  -- $DIRS /tmp/p_sq_$TAG /tmp/p_mfma_$TAG /tmp/p_tcc_$TAG /tmp/p_tex_$TAG > $OUT/pmc_$TAG.txt 2>&1
cd $R
ls -la $OUT
