#!/bin/bash
# The ONE runner of a round's gpurun calls (from the repo root):   gpurun --timeout T -- 'bash tools/gpu_round.sh <stage>...'
#   suite      python -m pytest tests -m gpu                      -> gpurun_out/round/pytest_gpu.log
#   smoke      __graft_entry__.smoke()                            -> .../smoke.log
#   bench      the driver's form (bench.py --gpus 1 --steps 20 --warmup 5): every leg's line + the record, full records
#   legs       every batch workload alone (maicity, kitti, kitti-large, ncd-incre) with its own cpu_baseline
#   profiles   tools/collect_profiles.sh for maicity / kitti / kitti-large (kernel stats, timelines, PMC -> JSON bench.py reads)
#   tiera      tools/tier_a_bench.py + tools/tier_a_hostcost.py (Tier A iteration times, statement by statement)
#   allowance  the trajectory test alone with -s: the allowance it printed -> .../trajectory_allowance.txt
# Copy what should be judged from gpurun_out/round/ to profiles/r0N_*.  (The one-off scripts of rounds 3-6 are under
# tools/archive/: each is one gpurun call of its round, kept for the lab books that cite them.)
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/round; mkdir -p $O
export SHINE_WORKLOAD_CACHE=${SHINE_WORKLOAD_CACHE:-/tmp/shine_wl_cache}
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for stage in "$@"; do
  case $stage in
    suite) ( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log ;;
    smoke) timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log ;;
    bench) ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-record-dir $O/bench_records ) \
             > $O/bench_default_driver_form.json.log 2> $O/bench_default_driver_form.err
           tail -4 $O/bench_default_driver_form.err | grep real; wc -c $O/bench_default_driver_form.json.log
           cut -c1-400 $O/bench_default_driver_form.json.log ;;
    legs)  for w in maicity kitti ncd-incre; do
             timeout 900 python bench.py --workload $w --no-extra-configs --full-record-dir $O/leg_records > $O/bench_$w.json.log 2> $O/bench_$w.err
           done
           timeout 900 python bench.py --workload kitti-large --no-extra-configs --no-cpu-baseline --full-record-dir $O/leg_records \
             > $O/bench_kitti-large.json.log 2> $O/bench_kitti-large.err
           cat $O/bench_maicity.json.log $O/bench_kitti.json.log $O/bench_kitti-large.json.log $O/bench_ncd-incre.json.log | cut -c1-300 ;;
    profiles) for spec in "maicity 262144 4" "kitti 1048576 3" "kitti-large 1048576 3"; do
             ( time timeout 2400 bash tools/collect_profiles.sh $spec ) > $O/collect_${spec%% *}.log 2>&1; tail -3 $O/collect_${spec%% *}.log
           done; cp gpurun_out/prof/* $O/ 2>/dev/null ;;
    tiera) timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -8
           for m in maicity incre eik; do timeout 300 python tools/tier_a_hostcost.py $m > $O/tier_a_hostcost_$m.log 2>&1; grep -v amdgpu $O/tier_a_hostcost_$m.log | head -16; done ;;
    allowance) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "incremental_trajectory_at_config_4_shape" > $O/trajectory.log 2>&1
           grep -i "allowance\|beyond\|passed\|failed" $O/trajectory.log | tee $O/trajectory_allowance.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
