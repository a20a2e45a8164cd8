#!/bin/bash
# gpurun call: Tier A tests + iteration times of the reference's loop body on this package's nodes
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "tier_a or autograd or fused_mlp or drop_in" > $O/pytest_tiera.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tiera.log; tail -15 $O/pytest_tiera.log
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -6
