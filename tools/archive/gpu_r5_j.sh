#!/bin/bash
# round 5, call J: work-balanced tile ranges (shine_balance_tiles) — kernel alone on equal shares against balanced shares, far and
# near builds, on kitti-large and on the cache-resident kitti map; the far-build parity test (now balanced by default)
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r05j; mkdir -p $O
AB_BALANCE=1 AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py tools/ab/lib_pk.so@6,5 > $O/ab_far_balance.txt 2>&1; grep -v "^$" $O/ab_far_balance.txt | grep -v amdgpu | tail -6
AB_BALANCE=1 timeout 600 python tools/ab_build.py tools/ab/lib_pk.so@6 > $O/ab_near_balance.txt 2>&1; grep -v "^$" $O/ab_near_balance.txt | grep -v amdgpu | tail -6
timeout 900 python -m pytest tests -q -m gpu -k "far_build or pool_mode_step or weighted_bce or sharded" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sub.log
timeout 600 python bench.py --workload kitti-large --no-extra-configs --no-cpu-baseline > $O/bench_kitti-large.json.log 2> $O/bench_kitti-large.err; python - <<'PY'
import json
for l in open("gpurun_out/r05j/bench_kitti-large.json.log"):
    if l.startswith("{"):
        r=json.loads(l); rf=r["roofline"]; print("kitti-large: %.4g samples/s, %.4f ms/step, kernel %.4f ms, %s frac %.3f, geometry %s" % (r["value"], r["ms_per_step"], rf["kernel_ms"], rf["bound"], rf["frac"], rf["launch_geometry"]))
PY
