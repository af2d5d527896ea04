#!/bin/bash
# round 5, twenty-fifth GPU session: velocity streams confined to n compute units (LTPL_VEL_CUS), same-box A/B
export TMPDIR=/tmp
T=${R05TAG:-r05A}
mkdir -p gpurun_out/$T
timeout 900 tools/vel_cus_ab.sh 0 16 32 64 > gpurun_out/$T/vel_cus_ab.txt 2>&1; cat gpurun_out/$T/vel_cus_ab.txt
