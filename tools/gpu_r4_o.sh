#!/bin/bash
# round 4, call o: the L2's fp32 atomic rate for the scatter's shape
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$R/gpurun_out/r04o; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics_rows tools/ubench/atomics_rows.hip 2>/dev/null
timeout 120 /tmp/atomics_rows > $O/ubench_atomics_rows.txt 2>&1; cat $O/ubench_atomics_rows.txt
