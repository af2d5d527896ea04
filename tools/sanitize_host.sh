#!/bin/bash
# Host-side memory / UB check (no GPU needed): builds the oracle and the host-logic harness -- i.e. the product's planner state machine
# csrc/fleet_core.hpp through csrc/planner_host.hpp (+ csrc/planner_core.hpp) behind the oracle's arithmetic -- with AddressSanitizer + UndefinedBehaviorSanitizer,
# runs the CPU test suite through them (closed-loop replays of every recording, all tracks), and restores the normal builds.
# Found in round 2: a reference into a vector kept across push_back (emergency trajectory), memcpy from an empty vector's null data().
#   tools/sanitize_host.sh [pytest args]            default: the whole CPU suite
set -u
cd "$(dirname "$0")/.."
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
make -s -B -C oracle CFLAGS="-O1 -fPIC -std=c11 -ffp-contract=off $SAN" CXXFLAGS="-O1 -fPIC -std=c++17 -ffp-contract=off $SAN" || exit 1
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tests/ --deselect tests/test_abi_symbols.py)
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 \
  python -m pytest "${ARGS[@]}" -q -s -m "not gpu" -p no:cacheprovider > /tmp/ltpl_sanitize.log 2>&1
rc=$?
make -s -B -C oracle || exit 1                      # back to the normal builds
n=$(grep -c "runtime error\|ERROR: AddressSanitizer" /tmp/ltpl_sanitize.log)
tail -1 /tmp/ltpl_sanitize.log
echo "sanitizer reports: $n (log: /tmp/ltpl_sanitize.log), pytest exit code $rc"
[ "$rc" -eq 0 ] && [ "$n" -eq 0 ]
