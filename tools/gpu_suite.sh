#!/bin/bash
# gpurun call: the whole GPU suite + the default bench line + N = 4096
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_maicity_nocpu.json 2> $O/bench_maicity_nocpu.err; cut -c1-900 $O/bench_maicity_nocpu.json
timeout 600 python bench.py --no-cpu-baseline --points 4096 > $O/bench_maicity_4096.json 2> $O/bench_maicity_4096.err; cut -c1-400 $O/bench_maicity_4096.json; grep -o '"kernel_ms": [0-9.]*' $O/bench_maicity_4096.json
