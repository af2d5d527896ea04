#!/bin/bash
# the switches used here only exist in the experiment build of the library (include/ltpl_hip.h)
export LTPL_HIP_LIB="$(cd "$(dirname "$0")/.." && pwd)/graphbasedlocaltrajectoryplanner_amd/csrc/libltpl_hip_exp.so"
# instruction counts of the path kernel per phase, by ablation (LTPL_ABLATE) + SQ instruction counters
# (--no-extra is essential: without it every pass runs the whole default bench -- fleets, C3, C5 -- under the counters: 16 GPU-minutes in round 4)
for A in ${ABL:-0 1 4 6 7}; do
  LTPL_ABLATE=$A LTPL_NO_OVERLAP=1 tools/pmc_pass.sh abl$A "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" --batch 8192 --steps 10 --warmup 2 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra --exact-steps 2>/dev/null | awk -v a=$A '/k_paths/{f=1;next} /^[a-z_]/{f=0} f{printf "ablate=%s %s\n", a, $0}'
done
