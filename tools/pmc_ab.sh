#!/bin/bash
# Counter passes over a few launches of the batch path kernel (alone: tools/dbg_plain_paths.py) for several library variants on ONE box:
#   [PMC_WORKLOAD=c2|c3] [PMC_N=<scenarios>] [PMC_TRAFFIC=0] [PMC_TAG=r05x] tools/pmc_ab.sh <lib|base> ...
# per variant: two SQ passes (the SQ block holds ~8 counters at a time) and -- unless PMC_TRAFFIC=0 -- the two HBM passes (FETCH_SIZE,
# WRITE_SIZE, each in its own run as MI355X_MICROARCH.md prescribes). tools/pmc_summarise.py prints the per-scenario figures and writes
#   gpurun_out/pmc_issue_<variant>[_c3].json, gpurun_out/pmc_traffic_<variant>[_c3].json
# = what bench.py reads as profiles/pmc_issue[_c3].json / profiles/pmc_traffic[_c3].json, each with the BUILD STAMP of the profiled
# library (digest of the path kernel's instruction stream + register metadata, __graft_entry__.build_stamp).
export TMPDIR=/tmp
WL=${PMC_WORKLOAD:-c2}
if [ "$WL" = c3 ]; then N=${PMC_N:-8192}; else N=${PMC_N:-32768}; fi
for V in "$@"; do
  if [ "$V" = base ]; then unset LTPL_HIP_LIB; T=base; LIB=graphbasedlocaltrajectoryplanner_amd/csrc/libltpl_hip.so
  else export LTPL_HIP_LIB=$PWD/$V; T=$(basename $V .so); LIB=$V; fi
  PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT")
  if [ "${PMC_TRAFFIC:-1}" != 0 ]; then PASSES+=("FETCH_SIZE" "WRITE_SIZE"); fi
  if [ -n "${PMC_EXTRA:-}" ]; then PASSES+=("$PMC_EXTRA"); fi
  for PASS in "${PASSES[@]}"; do
    OUT=gpurun_out/pmcab_${T}_${WL}_$(echo $PASS | tr ' ' '_' | cut -c1-24); rm -rf $OUT; mkdir -p $OUT
    timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT -o p -- python tools/dbg_plain_paths.py $N $WL 4 > $OUT/run.log 2>&1
    python tools/pmc_summarise.py "$OUT/p_counter_collection.csv" "$T" "$WL" "$LIB" || tail -5 $OUT/run.log
  done
done
