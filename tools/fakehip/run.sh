#!/bin/bash
# Host code of libltpl_hip.so under ASan + UBSan without a GPU: builds the library against the stand-in runtime (build.sh) and drives
# every entry point (exercise.py). Exit code 0 = the driver finished and the sanitizers reported nothing.
set -u
cd "$(dirname "$0")"
./build.sh > /dev/null || exit 1
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
[ -f "$RT/libclang_rt.asan-x86_64.so" ] || RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
LTPL_NO_SELFTEST=1 LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1 python exercise.py "$@" > /tmp/ltpl_fakehip.log 2>&1
rc=$?
n=$(grep -c ": runtime error:\|ERROR: AddressSanitizer" /tmp/ltpl_fakehip.log)
grep -v "^    (" /tmp/ltpl_fakehip.log | tail -8
echo "sanitizer reports: $n (log: /tmp/ltpl_fakehip.log), driver exit code $rc"
[ "$rc" -eq 0 ] && [ "$n" -eq 0 ]
