#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 600 python tools/tier_a_bench.py > $O/tier_a_bench.log 2>&1; grep -v amdgpu $O/tier_a_bench.log | tail -4
timeout 600 python bench.py --workload ncd-incre --no-cpu-baseline --no-extra-configs --unroll 10 > $O/bench_ncd.json.log 2> $O/bench_ncd.err
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json.log 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_ncd", "bench_default"):
    for l in open("gpurun_out/r04f/%s.json.log" % f):
        if l.startswith("{"):
            r = json.loads(l)
            print(f, "%.4g %s" % (r["value"], r["unit"]), "%.4f ms/step" % r["ms_per_step"], r.get("frames_per_s"), r.get("per_frame_ms_median"),
                  (r["config"].get("window_ms") or {}).get("all"), "plan", r["config"].get("pool_plan_ms"), (r.get("like_for_like") or {}).get("gpu"))
            for k, v in (r.get("configs") or {}).items():
                print("   ", k, v.get("value"), v.get("ms_per_step"), v.get("frames_per_s"), v.get("per_frame_ms_median"), v.get("error"))
PY
AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300 timeout 900 python tools/ab_build.py shine_mapping_amd/lib/libshine_check.so tools/ab/lib_gb8.so tools/ab/lib_gb2.so > $O/ab_gather_batch_kitti_large.txt 2>&1; grep -v "^$" $O/ab_gather_batch_kitti_large.txt | tail -5
