#!/usr/bin/env python
"""Per-launch PMC figures of one kernel from several rocprofv3 --pmc runs (one db directory per counter group) -> the
JSON record bench.py reads (profiles/r03_pmc_<workload>_<points>_L<levels>.json).

    python tools/pmc_to_json.py --kernel k_step_v --out profiles/r03_pmc_maicity_262144_L4.json \
        --meta workload=maicity points=262144 levels=4 -- /tmp/p_FETCH /tmp/p_WRITE /tmp/p_sq /tmp/p_mfma

Every counter is summed over its instances (XCDs / SEs) per dispatch, then averaged over the dispatches of the kernel
whose grid matches the most frequent grid size (the bench's own launches).  Corrections (MI355X_MICROARCH.md §HBM):
FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled.
"""
import argparse
import glob
import json
import os
import sqlite3
import sys
from collections import Counter, defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def read_db(root, pat):
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not dbs:
        raise SystemExit("no *_results.db under %s" % root)
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, d.event_id, d.grid_size_x, i.name, sum(e.value), d.end - d.start, count(e.value) from %s e "
         "join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "group by d.event_id, i.name" % (pmc, info, kd, ks))
    per = defaultdict(dict)
    inst = {}
    grids, durs, name = {}, {}, None
    for kname, ev, grid, cname, val, dur, cnt in cur.execute(q):
        if pat not in kname:
            continue
        name = kname.replace(".kd", "")
        per[ev][cname] = val
        inst[cname] = cnt  # hardware instances (XCDs / SEs / CUs ...) the counter was summed over
        grids[ev] = grid
        durs[ev] = dur
    if not per:
        raise SystemExit("kernel %s not found in %s" % (pat, dbs[0]))
    top = Counter(grids.values()).most_common(1)[0][0]
    keep = [ev for ev in per if grids[ev] == top]
    out = {}
    for c in sorted({c for ev in keep for c in per[ev]}):
        vals = [per[ev][c] for ev in keep if c in per[ev]]
        out[c] = sum(vals) / len(vals)
    return name, out, len(keep), sum(durs[ev] for ev in keep) / len(keep) / 1e3, top, inst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_step_v")
    ap.add_argument("--out", required=True)
    ap.add_argument("--meta", nargs="*", default=[])
    ap.add_argument("--command", default="")
    ap.add_argument("--lib", default=None, help="library whose kernel code is hashed into the record (default: the in-tree .so)")
    ap.add_argument("--valu-cycles", type=float, default=2.5,
                    help="fp32-datapath cycles per plain VALU wave-instruction (profiles/r03_ubench_calibration.txt)")
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    counters, launches, name, dur_us, grid, instances = {}, {}, None, {}, None, {}
    for d in a.dirs:
        name, c, n, us, grid, inst = read_db(d, a.kernel)
        for k, v in c.items():
            counters[k] = v
            launches[k] = n
            dur_us[k] = us
            instances[k] = inst.get(k)
    rec = {"kernel": name, "grid_size_x": grid, "launches_averaged": launches, "counters_per_launch": counters,
           "instances_summed": instances,
           "kernel_us_under_profiler": dur_us, "command": a.command,
           "source": "rocprofv3 --kernel-trace --pmc <group> (one run per group), tools/collect_profiles.sh + "
                     "tools/pmc_to_json.py"}
    # tie the counters to the code they were collected on: bench.py drops the record when the loaded library holds
    # another build of this kernel (tools/kernel_hash.py)
    try:
        from kernel_hash import kernel_code_sha256

        rec["kernel_code_sha256"] = kernel_code_sha256(name, a.lib)
    except Exception as e:  # pragma: no cover
        rec["kernel_code_sha256"] = None
        rec["kernel_code_sha256_error"] = str(e)
    if counters.get("SQ_WAVE_CYCLES", 0) > 0:
        wc = counters["SQ_WAVE_CYCLES"]
        rec["wave_cycle_split"] = {
            "waiting_at_waitcnt_or_barrier": counters.get("SQ_WAIT_ANY", 0.0) / wc,
            "issue_stalled": counters.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            "issuing": counters.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
            "formula": "SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, guide §PMC slots)"}
    for kv in a.meta:
        k, v = kv.split("=", 1)
        rec[k] = int(v) if v.isdigit() else v
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        rec["FETCH_SIZE_KB_per_launch"] = counters["FETCH_SIZE"]
        rec["WRITE_SIZE_KB_per_launch"] = counters["WRITE_SIZE"]
        rec["hbm_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
        rec["correction"] = ("MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced "
                             "reads -> doubled; WRITE_SIZE as reported; both in KB (x1024)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and counters.get("SQ_BUSY_CYCLES", 0) > 0:
        # SQ_BUSY_CYCLES is summed over its instances (shader engines); one instance's value is the kernel's duration in
        # shader cycles.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles of every SIMD (= 32 x the 16x16x4 fp32 MFMAs issued).
        n_simd = 256 * 4
        kcycles = counters["SQ_BUSY_CYCLES"] / max(instances.get("SQ_BUSY_CYCLES") or 1, 1)
        busy = counters["SQ_VALU_MFMA_BUSY_CYCLES"]
        rec["kernel_shader_cycles"] = kcycles
        rec["mfma_busy_cycles_per_launch"] = busy
        rec["mfma_util"] = busy / (kcycles * n_simd)
        rec["mfma_util_formula"] = ("SQ_VALU_MFMA_BUSY_CYCLES / ((SQ_BUSY_CYCLES / its instances) * 1024 SIMDs): the fraction "
                                    "of SIMD-cycles the matrix pipe is busy")
        if "SQ_INSTS_VALU" in counters:
            # Exact-fp32 MFMA runs on the SIMD's fp32 FMA lanes and does not overlap VALU work of another wave
            # (tools/ubench/mfma_valu_overlap.hip), so MFMA busy cycles and the plain VALU instructions' cycles add up to
            # the occupancy of ONE datapath.  Cycles per VALU instruction: calibrated on the micro-benchmark
            # (tools/calibrate_datapath.py -> profiles/r03_ubench_calibration.txt), not the 4 of the r02 record.
            valu = a.valu_cycles * (counters["SQ_INSTS_VALU"] - counters.get("SQ_INSTS_MFMA", 0.0))
            rec["valu_busy_cycles_per_launch"] = valu
            rec["valu_cycles_per_instruction"] = a.valu_cycles
            rec["fp32_datapath_util"] = (busy + valu) / (kcycles * n_simd)
            rec["fp32_datapath_util_formula"] = ("(SQ_VALU_MFMA_BUSY_CYCLES + %.2f * (SQ_INSTS_VALU - SQ_INSTS_MFMA)) / "
                                                 "(kernel shader cycles * 1024 SIMDs)" % a.valu_cycles)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
