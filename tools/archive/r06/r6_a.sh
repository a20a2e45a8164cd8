#!/bin/bash
# round 6, call A: the new config-5-shape test, the driver-form bench (compact record), then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_scale_parity.py -m gpu -q -x -s -k "config_5_shape" ) > $O/pytest_config5.log 2>&1; tail -8 $O/pytest_config5.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err; tail -5 $O/bench_default_driver_form.err
wc -c $O/bench_default_driver_form.json.log; cut -c1-600 $O/bench_default_driver_form.json.log
cp -r gpurun_out/bench_records $O/ 2>/dev/null
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
