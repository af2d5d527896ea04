#!/bin/bash
# Builds tools/fakehip/build/libltpl_hip_fake.so: the HOST-ONLY object of csrc/ltpl_hip.hip (no device code) with ASan + UBSan, linked
# against the stand-in runtime fakehip.cpp instead of libamdhip64. See fakehip.cpp for what this can and cannot show.
set -eu
cd "$(dirname "$0")"
ROOT=$(cd ../.. && pwd)
mkdir -p build
SAN="-fsanitize=${FAKEHIP_SAN:-address,undefined} -fno-omit-frame-pointer -g -O1"      # FAKEHIP_SAN=thread: ThreadSanitizer build
CLANG=/opt/rocm/lib/llvm/bin/clang++          # one toolchain for all objects: the sanitizer run-time is clang's (see run.sh)
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-host-only -c $SAN -std=c++17 -ffp-contract=off -fPIC -D_GLIBCXX_ASSERTIONS \
    -Wno-unused-function -o build/ltpl_host.o "$ROOT/graphbasedlocaltrajectoryplanner_amd/csrc/ltpl_hip.hip"
$CLANG -c $SAN -std=c++17 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o build/fakehip.o fakehip.cpp
# the host stubs reference the embedded code object by a per-build symbol: give it a dummy definition
FATBIN=$(nm -u build/ltpl_host.o | awk '/__hip_fatbin/ {print $2}')
echo "char $FATBIN[16] = {0};" > build/fatbin_dummy.c
gcc -c -fPIC -o build/fatbin_dummy.o build/fatbin_dummy.c
$CLANG -shared -shared-libsan $SAN -o build/libltpl_hip_fake.so build/ltpl_host.o build/fakehip.o build/fatbin_dummy.o
echo "$PWD/build/libltpl_hip_fake.so"
