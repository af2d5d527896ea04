#!/bin/bash
# Builds tools/fakehip/build/libltpl_hip_fake.so: the HOST-ONLY object of csrc/ltpl_hip.hip (no device code) with ASan + UBSan, linked
# against the stand-in runtime fakehip.cpp instead of libamdhip64. See fakehip.cpp for what this can and cannot show.
set -eu
cd "$(dirname "$0")"
ROOT=$(cd ../.. && pwd)
# FAKEHIP_SAN=thread: ThreadSanitizer build; FAKEHIP_SAN=none: plain -O3 build into build_plain/ (host-only timing, address-space-limit test)
if [ "${FAKEHIP_SAN:-}" = "none" ]; then OUT=build_plain; SAN="-O3"; LSAN=""
else OUT=build; SAN="-fsanitize=${FAKEHIP_SAN:-address,undefined} -fno-omit-frame-pointer -g -O1"; LSAN="-shared-libsan"; fi
mkdir -p $OUT
CLANG=/opt/rocm/lib/llvm/bin/clang++          # one toolchain for all objects: the sanitizer run-time is clang's (see run.sh)
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-host-only -c $SAN -std=c++17 -ffp-contract=off -fPIC -D_GLIBCXX_ASSERTIONS \
    -Wno-unused-function -o $OUT/ltpl_host.o "$ROOT/graphbasedlocaltrajectoryplanner_amd/csrc/ltpl_hip.hip"
$CLANG -c $SAN -std=c++17 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $OUT/fakehip.o fakehip.cpp
# the host stubs reference the embedded code object by a per-build symbol: give it a dummy definition
FATBIN=$(nm -u $OUT/ltpl_host.o | awk '/__hip_fatbin/ {print $2}')
echo "char $FATBIN[16] = {0};" > $OUT/fatbin_dummy.c
gcc -c -fPIC -o $OUT/fatbin_dummy.o $OUT/fatbin_dummy.c
$CLANG -shared $LSAN $SAN -o $OUT/libltpl_hip_fake.so $OUT/ltpl_host.o $OUT/fakehip.o $OUT/fatbin_dummy.o
echo "$PWD/$OUT/libltpl_hip_fake.so"
