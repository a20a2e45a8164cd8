#!/bin/bash
# round 6, call O: `python bench.py` with no flags (defaults: N = 1, K = 200, W = 20) — wall time and the record
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06o; mkdir -p $O
( time timeout 1200 python bench.py ) > $O/bench_noflags.json.log 2> $O/bench_noflags.err; tail -4 $O/bench_noflags.err | grep real; wc -c $O/bench_noflags.json.log; tail -1 $O/bench_noflags.json.log
