#!/bin/bash
# round 5, call K: where the HOST time of an incremental frame goes (cProfile over bench.py's ncd-incre loop, Tier B)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05k; mkdir -p $O
timeout 600 python -m cProfile -o $O/ncd.prof bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --no-tier-a --steps 40 --warmup 5 > $O/bench.json.log 2> $O/bench.err
python - <<'PY'
import pstats
st = pstats.Stats("gpurun_out/r05k/ncd.prof")
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats("shine_mapping_amd|bench.py", 45)
PY
