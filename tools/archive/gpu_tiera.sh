#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -8
