#!/bin/bash
# round 5, call T: the bench configurations on SURVEY 8(d)'s full scan recipe (poses 1 m apart x 64 beams x 1800 azimuths), next to the default's thinner scans
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05t; mkdir -p $O
for w in kitti-recipe kitti; do
( time timeout 900 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 ) > $O/bench_$w.json.log 2> $O/err_$w.txt; tail -3 $O/err_$w.txt | grep real
done
python - <<'PY'
import json
for f in ("kitti-recipe", "kitti"):
    for l in open("gpurun_out/r05t/bench_%s.json.log" % f):
        if l.startswith("{"):
            r = json.loads(l); rf = r["roofline"]; c = r["config"]
            print(f, "value %.4g, %.4f ms/step, kernel %.4f ms, frac %.3f; pool %d samples, rows %s, table %.1f MB, pool plan %.1f ms" % (
                r["value"], r["ms_per_step"], rf["kernel_ms"], rf["frac"], c["pool_samples"], c["corner_rows"], c["feature_table_bytes"] / 1e6, c["pool_plan_ms"]))
PY
grep -i "error\|Traceback" $O/err_*.txt | head -5; rocm-smi --showmeminfo vram 2>/dev/null | grep -i used | head -2
