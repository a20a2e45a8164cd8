#!/bin/bash
# gpurun call C of round 3: per-role cycle counters of k_step_v5 under ablations (what is the gather / scatter / decoder
# wave's time made of), and the SQ counters of k_step_v5 next to k_step_v3's.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
for V in a0 a1 a4 a8 a2 a15; do
  echo "== ablation $V" >> $O/v5_prof_abl.txt
  timeout 300 python tools/v5_prof.py tools/ab/lib_$V.so maicity:4 >> $O/v5_prof_abl.txt 2>&1
done
grep -v amdgpu.ids $O/v5_prof_abl.txt
cd /tmp && export TMPDIR=/tmp
for K in v3 v5; do
  ENVV=""; [ $K = v5 ] && ENVV="SHINE_V5_MIN_POINTS=0"
  BENCH="python $R/tools/run_with_lib.py $R/tools/ab/lib_pch4dma.so bench.py --workload maicity --no-cpu-baseline --no-extra-configs --steps 6 --warmup 2 --no-graph"
  env $ENVV rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/q_sq_$K -o run -- $BENCH > $O/pmc_sq_$K.log 2>&1
  env $ENVV rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d /tmp/q_mf_$K -o run -- $BENCH > $O/pmc_mf_$K.log 2>&1
  python $R/tools/pmc_to_json.py --kernel k_step_v --out $O/pmc_cmp_$K.json --lib $R/tools/ab/lib_pch4dma.so -- /tmp/q_sq_$K /tmp/q_mf_$K > $O/pmc_cmp_$K.txt 2>&1
  python - <<PY
import json
d=json.load(open("$O/pmc_cmp_$K.json"))
c=d["counters_per_launch"]; t=262144/16
print("$K", d["kernel"][:40], {k: round(v/t,1) for k,v in sorted(c.items()) if k.startswith("SQ_")}, "cycles", round(d.get("kernel_shader_cycles",0)), d.get("wave_cycle_split"))
PY
done
