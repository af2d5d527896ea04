#!/bin/bash
# A/B of library variants on ONE box, fleet tape rate: tools/ab_fleet.sh <lib or "base"> ...   (alternating, 2 rounds; --mix with MIX=1)
for round in 1 2; do
  for V in "$@"; do
    if [ "$V" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$V; fi
    printf '%-40s ' "$V"; python tools/fleet_rate.py --ticks 100 --reps 2 ${MIX:+--mix} 2>&1 | grep closed_loop_device_ticks_per_s
  done
done
