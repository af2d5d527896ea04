#!/bin/bash
# the switches used here only exist in the experiment build of the library (include/ltpl_hip.h)
export LTPL_HIP_LIB="$(cd "$(dirname "$0")/.." && pwd)/graphbasedlocaltrajectoryplanner_amd/csrc/libltpl_hip_exp.so"
# Run the C3 parity probe with the team's LDS (LTPL_LDS_POISON) or the stream's scratch arena (LTPL_SCRATCH_POISON)
# pre-filled with different words. usage: poison_sweep.sh lds|scratch|vgpr <word> ...
kind=$1; shift
for p in "$@"; do
  echo "== $kind poison $p"
  if [ "$kind" = lds ]; then export LTPL_LDS_POISON=$p; elif [ "$kind" = vgpr ]; then export LTPL_VGPR_POISON=$p; else export LTPL_SCRATCH_POISON=$p; fi
  timeout 100 python tools/dbg_c3.py 2>&1 | grep -a "field\|node seq\|ALL EQ" | cut -c1-160
done
