#!/bin/bash
# ONE parameterised GPU-box session (replaces the one-shot tools/r05_call*.sh of round 5; their outputs stay cited in docs/HISTORY.md):
#
#   gpurun --timeout 2400 -- 'tools/gpu_session.sh <tag> <step> [<step> ...]'
#
# Every step writes under gpurun_out/<tag>/ (merged back into the build container; what is to be judged is copied to profiles/<tag>_*).
# Steps (executed in the order given; a step that fails is reported and the session goes on):
#   tests[:<pytest -k expr>]   the GPU parity suite (pytest -m gpu), optionally a subset             -> gputest.txt
#   smoke                      __graft_entry__.smoke()                                                -> smoke.txt
#   bench[:<args>]             python bench.py [args] (default: the driver's form, no flags)          -> bench.json, bench_time.txt
#   quick                      short bench line without CPU leg / extras                              -> quick.json
#   stats                      rocprofv3 --kernel-trace --stats of a short bench run                  -> kernel_stats.csv
#   stats_alone                the same with LTPL_NO_OVERLAP=1 (per-kernel durations without overlap) -> kernel_stats_alone.csv
#   pmc / pmc_c3               stamped SQ + HBM counter passes over the batch path kernel (tools/pmc_ab.sh base)
#   fleet_stats                kernel stats of a short mixed fleet tape (32 768 planners)             -> fleet_kernel_stats.csv
#   ab:<lib>,<lib>,...         alternating same-box A/B of library variants (tools/ab_bench.sh; "base" = the shipped library)
#   abfleet:<lib>,...          the same on the mixed fleet tape (tools/ab_fleet.sh)
#   tick:<env>|<env>|...       alternating A/B of the single-tick latency under environment settings (tools/tick_ab.sh; "-" = default)
#   c3 / c5                    C3 throughput (tools/c3_rate.py) / C5 latency (tools/c5_latency.py 300 300)
#   run:<command>              any other command, output -> run_<n>.txt
export TMPDIR=/tmp
T=${1:?tag}; shift
O=gpurun_out/$T; mkdir -p $O
n_run=0
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [ "$STEP" != "$NAME" ] && ARG=${STEP#*:}
  echo "=== [$T] $NAME $ARG"
  case $NAME in
    tests)
      if [ -n "$ARG" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$ARG" > $O/gputest.txt 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; fi
      echo "gpu tests rc=$?"; tail -6 $O/gputest.txt ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt ;;
    bench)
      ( time timeout 1200 python bench.py $ARG > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"
      python tools/bench_brief.py $O/bench.json; tail -3 $O/bench_time.txt; tail -3 $O/bench.err ;;
    quick)
      timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu --latency-ticks 300 --dropin-ticks 600 --no-extra $ARG > $O/quick.json 2> $O/quick.err; echo "quick rc=$?"
      python tools/bench_brief.py $O/quick.json ;;
    stats|stats_alone)
      [ $NAME = stats_alone ] && export LTPL_NO_OVERLAP=1
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$NAME -o k -- python bench.py --steps 50 --warmup 5 --no-cpu --latency-ticks 200 --dropin-ticks 300 --exact-steps --no-extra > $O/$NAME.log 2>&1; echo "$NAME rc=$?"
      unset LTPL_NO_OVERLAP
      F=kernel_stats.csv; [ $NAME = stats_alone ] && F=kernel_stats_alone.csv
      find $O/$NAME -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/$F; head -14 $O/$F | cut -c1-170 ;;
    pmc)
      PMC_TAG=$T timeout 1200 tools/pmc_ab.sh base > $O/pmc_c2.txt 2>&1; cat $O/pmc_c2.txt ;;
    pmc_c3)
      PMC_TAG=$T PMC_WORKLOAD=c3 PMC_N=32768 timeout 1200 tools/pmc_ab.sh base > $O/pmc_c3.txt 2>&1; cat $O/pmc_c3.txt ;;
    fleet_stats)
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fleet_stats -o k -- python tools/fleet_rate.py --planners 32768 --ticks 50 --mix --reps 1 > $O/fleet_stats.log 2>&1; echo "fleet stats rc=$?"
      find $O/fleet_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fleet_kernel_stats.csv; head -18 $O/fleet_kernel_stats.csv | cut -c1-150 ;;
    ab)
      timeout 1500 tools/ab_bench.sh $(echo $ARG | tr ',' ' ') > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt ;;
    abfleet)
      MIX=1 timeout 1500 tools/ab_fleet.sh $(echo $ARG | tr ',' ' ') > $O/ab_fleet.txt 2>&1; cat $O/ab_fleet.txt ;;
    tick)
      IFS='|' read -ra ENVS <<< "$ARG"
      for i in "${!ENVS[@]}"; do [ "${ENVS[$i]}" = "-" ] && ENVS[$i]="LTPL_AB_DEFAULT=1"; done      # ("-" = the default settings)
      timeout 1200 tools/tick_ab.sh "${ENVS[@]}" > $O/tick_ab.txt 2>&1; cat $O/tick_ab.txt ;;
    c3)
      timeout 600 python tools/c3_rate.py $ARG > $O/c3.txt 2>&1; cat $O/c3.txt ;;
    c5)
      timeout 600 python tools/c5_latency.py 300 300 > $O/c5.txt 2>&1; tail -3 $O/c5.txt ;;
    run)
      n_run=$((n_run + 1)); timeout 1500 bash -c "$ARG" > $O/run_$n_run.txt 2>&1; echo "run rc=$?"; tail -40 $O/run_$n_run.txt ;;
    *) echo "unknown step $NAME" ;;
  esac
done
