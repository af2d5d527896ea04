#!/bin/bash
# the switches used here only exist in the experiment build of the library (include/ltpl_hip.h)
export LTPL_HIP_LIB="$(cd "$(dirname "$0")/.." && pwd)/graphbasedlocaltrajectoryplanner_amd/csrc/libltpl_hip_exp.so"
# instruction counts of the path kernel per phase, by ablation (LTPL_ABLATE) + SQ instruction counters
for A in 0 1 4 6 7; do
  LTPL_ABLATE=$A LTPL_NO_OVERLAP=1 tools/pmc_pass.sh abl$A "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" --batch 8192 --steps 10 --warmup 2 --no-cpu --latency-ticks 0 2>/dev/null | awk -v a=$A '/k_paths/{f=1;next} /^[a-z_]/{f=0} f{printf "ablate=%s %s %s\n", a, $1, $4}'
done
