#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06s; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 600 python tools/small_batch_geometry.py 2>&1 | grep -v amdgpu | tee $O/small_batch_geometry.txt
