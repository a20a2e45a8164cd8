#!/bin/bash
# gpurun call: in-process A/B of library variants (tools/mk_variant.py) + a quick parity run of the in-tree build
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r02
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_scale_parity.py tests/test_gpu_parity.py -m gpu -x -q -k "${AB_TESTS:-scale or golden or oracle}" > $O/ab_pytest.log 2>&1; tail -3 $O/ab_pytest.log
AB_ONLY=${AB_ONLY:-maicity:4} timeout 900 python tools/ab_build.py $AB_LIBS > $O/ab_result.txt 2>&1
tail -8 $O/ab_result.txt
