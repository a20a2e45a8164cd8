#!/bin/bash
# gpurun call 1 of round 4: the GPU suite of the refactored tree (step body as a device function, v5 removed, new tests), the
# grid-barrier micro-benchmark (design input of the persistent iteration kernel), and the fixed cost of the fused kernel at the
# reference's batch size: round-3 kernel vs the re-ordered prologue, per-wave phase counters.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
mkdir -p tools/ubench/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/grid_barrier tools/ubench/grid_barrier.hip 2>/dev/null
timeout 300 tools/ubench/bin/grid_barrier 200 > $O/grid_barrier.txt 2>&1; grep -c . $O/grid_barrier.txt; grep "256 threads,      0 words\|256 threads,    256 words" $O/grid_barrier.txt
for PTS in 4096 262144; do
  AB_POINTS=$PTS AB_PROF=1 AB_ONLY=maicity:4,maicity:3 timeout 600 python tools/ab_build.py tools/ab/lib_r03prof.so tools/ab/lib_r04prof.so > $O/ab_prologue_$PTS.txt 2>&1
  grep -v "^$" $O/ab_prologue_$PTS.txt | tail -12
done
