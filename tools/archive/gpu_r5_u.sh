#!/bin/bash
# round 5, call U: the default bench on SURVEY 8(d)'s full scan recipe; the thin-scan workloads of rounds 1-4 on the same box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05u; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json.log 2> $O/bench_default.err; tail -4 $O/bench_default.err | grep real
for w in maicity-thin kitti-thin; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-extra-configs > $O/bench_$w.json.log 2> $O/err_$w.txt
done
for w in maicity kitti; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-extra-configs > $O/bench_$w.json.log 2> $O/err_$w.txt
done
python - <<'PY'
import json
def show(tag, r):
    rf = r.get("roofline") or {}; c = r.get("config") or {}
    print("%-14s value %.4g, %.4f ms/step, kernel %s ms, frac %s (pmc %s); pool %s, rows %s" % (tag, r["value"], r["ms_per_step"], rf.get("kernel_ms"), rf.get("frac"), (rf.get("pmc") or {}).get("used"), c.get("pool_samples"), c.get("corner_rows")))
for l in open("gpurun_out/r05u/bench_default.json.log"):
    if l.startswith("{"):
        r = json.loads(l); show("default", r)
        for k, v in r["configs"].items():
            print("  ", k, "ERROR " + v["error"] if "error" in v else (v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("kernel_ms"), (v.get("roofline") or {}).get("frac"), v.get("frames_per_s"), (v.get("tier_a") or {}).get("frames_per_s")))
for w in ("maicity-thin", "maicity", "kitti-thin", "kitti"):
    for l in open("gpurun_out/r05u/bench_%s.json.log" % w):
        if l.startswith("{"):
            show(w, json.loads(l))
PY
grep -il "error\|Traceback" $O/*.err $O/err_*.txt
