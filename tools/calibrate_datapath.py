#!/usr/bin/env python
"""Calibrate the fp32-datapath occupancy formula on tools/ubench/mfma_valu_overlap (one rocprofv3 --pmc database per
mode).  Known truths: modes 0 / 3 issue only v_mfma_f32_16x16x4_f32 from one / two waves per SIMD (occupancy 1.0 of the
datapath: 32 cycles each, back to back), modes 1 / 4 issue only independent v_fma_f32 from one / two waves per SIMD.

    python tools/calibrate_datapath.py /tmp/cal_0 /tmp/cal_1 /tmp/cal_3 /tmp/cal_4

Prints, per mode, the counters per launch, the kernel's shader cycles, and the cycles per VALU instruction that make
   (SQ_VALU_MFMA_BUSY_CYCLES + w * (SQ_INSTS_VALU - SQ_INSTS_MFMA)) / (kernel shader cycles * 1024 SIMDs) = 1.
"""
import sys

from pmc_to_json import read_db

N_SIMD = 1024
rows = []
for d in sys.argv[1:]:
    name, c, n, us, grid, inst = read_db(d, "k")
    kcycles = c["SQ_BUSY_CYCLES"] / max(inst.get("SQ_BUSY_CYCLES") or 1, 1)
    valu = c.get("SQ_INSTS_VALU", 0.0) - c.get("SQ_INSTS_MFMA", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rows.append((d, c, kcycles, valu, busy, us))
    print("== %s  (%d launches, %.1f us under the profiler, grid %s)" % (d, n, us, grid))
    for k in sorted(c):
        print("   %-28s %16.0f   (summed over %s instances)" % (k, c[k], inst.get(k)))
    print("   kernel shader cycles %.0f  -> clock %.3f GHz" % (kcycles, kcycles / us / 1e3))
    print("   MFMA busy / (cycles x SIMDs)                      = %.3f" % (busy / (kcycles * N_SIMD)))
    if valu > 0:
        print("   plain VALU instructions per SIMD                  = %.0f" % (valu / N_SIMD))
        print("   cycles per VALU instruction for occupancy 1.0      = %.3f" % ((kcycles * N_SIMD - busy) / valu))
        print("   SQ_ACTIVE_INST_VALU x 4 / (cycles x SIMDs)        = %.3f" % (4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (kcycles * N_SIMD)))
