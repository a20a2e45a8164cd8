#!/bin/bash
# where do the waves of the fused kernel wait?  VMEM / LDS / SMEM occupancy-level counters (average latency = LEVEL / count)
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out/lat; O=$R/gpurun_out/lat
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
BENCH="python $R/bench.py --workload ${1:-maicity} --no-cpu-baseline --no-extra-configs --steps 6 --warmup 2 --no-graph --preheat-ms 0"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES -d /tmp/lat1 -o run -- $BENCH > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_ANY -d /tmp/lat2 -o run -- $BENCH > $O/p2.log 2>&1
python $R/tools/pmc_summary.py /tmp/lat1 k_step_v3 > $O/lat1.txt 2>&1
python $R/tools/pmc_summary.py /tmp/lat2 k_step_v3 > $O/lat2.txt 2>&1
cat $O/lat1.txt $O/lat2.txt | head -60; grep -c . $O/sq_counters.txt
