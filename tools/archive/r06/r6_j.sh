#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06u; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -x -k "record_pool_equals" ) > $O/pytest_new.log 2>&1; grep -v "^$" $O/pytest_new.log | tail -30
