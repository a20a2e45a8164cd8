#!/bin/bash
# round 6, call I: regulariser rider + ADVICE tests; Tier A host cost before / after (rider off / on)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06i; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests -m gpu -q -x -s -k "regulariser or cpp_nodes_hold or b3_gradient or plan or trajectory or tier_a" ) > $O/pytest_new.log 2>&1; grep -v "^$" $O/pytest_new.log | grep -v Warning | tail -30
for r in 0 1; do
SHINE_RIDER=$r timeout 300 python tools/tier_a_hostcost.py incre > $O/tier_a_hostcost_incre_rider$r.log 2>&1; grep -v amdgpu $O/tier_a_hostcost_incre_rider$r.log | head -14
done
