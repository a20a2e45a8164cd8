#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -x -s -k "config_5_shape" ) > $O/pytest_config5.log 2>&1; grep -v "^$" $O/pytest_config5.log | tail -30
