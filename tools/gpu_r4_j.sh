#!/bin/bash
# round 4, call j: where the HOST time of an incremental frame goes (cProfile over the bench's frame loop)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04j; mkdir -p $O
timeout 900 python -m cProfile -o /tmp/ncd.prof bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --steps 40 > $O/bench_prof.json.log 2> $O/bench_prof.err
python - > $O/host_profile.txt 2>&1 <<'PY'
import pstats
p = pstats.Stats("/tmp/ncd.prof")
p.sort_stats("tottime").print_stats(55)
p.sort_stats("cumulative").print_stats("shine_mapping_amd|bench.py", 70)
PY
head -150 $O/host_profile.txt | cut -c1-180
