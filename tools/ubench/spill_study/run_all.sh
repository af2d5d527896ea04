#!/bin/bash
# GPU box: every variant of the spill study; results -> gpurun_out/spill_study.txt
D=tools/ubench/spill_study
for L in default lib_spill lib_spill_O1 lib_calls_nospill lib_spill_calls; do
  if [ "$L" = "default" ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$D/$L.so; fi
  timeout 300 python $D/run.py 2048 2>&1 | tail -1
done | tee gpurun_out/spill_study.txt
