#!/bin/bash
# round 6, call K: the two-launch step (draw + zero-fill riding on the reduction launch) against the three-launch step, same box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
for w in maicity kitti kitti-large; do
 for mode in rider norider; do
  extra=""; [ $mode = norider ] && extra="--no-draw-rider"
  timeout 900 python bench.py --workload $w --no-extra-configs --no-cpu-baseline $extra --full-record-dir $O/records_$mode > $O/bench_${w}_$mode.json.log 2> $O/bench_${w}_$mode.err
  python - <<PY
import json
r=json.loads(open("$O/bench_${w}_$mode.json.log").read().strip().splitlines()[-1])
print("$w $mode", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["config"]["window_ms"], r["final_loss"])
PY
 done
done
