#!/bin/bash
# gpurun call: list the counters of this box, then texture-path (TA / TCP / TD) and instruction-mix PMC passes of the
# default bench kernel.  Usage: bash tools/gpu_pmc2.sh "<group1 counters>" "<group2 counters>" ...
cd $GRAFT_REPO_ROOT 2>/dev/null || true
R=$PWD
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_full.txt 2>&1
grep -o "^[[:space:]]*[A-Z][A-Za-z0-9_]*" $O/counters_full.txt | tr -d ' \t' | sort -u > $O/counters_names.txt
BENCH="python $R/bench.py --workload maicity --points 262144 --levels 4 --no-cpu-baseline --steps 6 --warmup 2 --no-graph"
i=0; DIRS=""
for G in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -d /tmp/pg_$i -o run -- $BENCH > $O/pmc2_group$i.log 2>&1
  DIRS="$DIRS /tmp/pg_$i"
done
python $R/tools/pmc_to_json.py --kernel k_step_v --out $O/pmc2.json -- $DIRS > $O/pmc2.txt 2>&1
tail -70 $O/pmc2.txt
wc -l $O/counters_names.txt
