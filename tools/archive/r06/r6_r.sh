#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06r; mkdir -p $O tools/ubench/bin
hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics_rows tools/ubench/atomics_rows.hip || exit 1
/tmp/atomics_rows > $O/ubench_atomics_rows_wide.txt 2>&1; cat $O/ubench_atomics_rows_wide.txt
