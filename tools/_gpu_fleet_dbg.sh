set -u
O=gpurun_out/fleet9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fleet.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit $?"; tail -5 $O/tests.log
timeout 300 python tools/fleet_rate.py --planners 8192 --ticks 200 --reps 1 > $O/rate_8192.log 2>&1; echo "rate exit $?"; tail -3 $O/rate_8192.log
timeout 300 python tools/fleet_rate.py --planners 8192 --ticks 200 --reps 1 --mix > $O/rate_8192_mix.log 2>&1; echo "rate mix exit $?"; tail -3 $O/rate_8192_mix.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fleet -- python $GRAFT_REPO_ROOT/tools/fleet_rate.py --planners 8192 --ticks 50 --reps 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof exit $?"
