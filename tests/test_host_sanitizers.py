"""CPU: the HOST code of libltpl_hip.so (argument checks, packing, buffer sizing, copy-in / copy-out) under AddressSanitizer + UBSan.

There is no GPU in the build container, so the library's host-only object is linked against a stand-in runtime (tools/fakehip:
"device" memory on the heap, copies = memcpy, launches do nothing) and every entry point of the C ABI is driven with real inputs; the
sanitizers watch both ends of every transfer. This is a memory-safety check of the host side and nothing else -- no device code runs
and no result is looked at (that is what the -m gpu tests are for). The planner's state machine has its own sanitizer run over the
closed-loop recordings (tools/sanitize_host.sh, ~4 min, not part of the suite)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"),
                    reason="clang's AddressSanitizer run-time is not installed")
def test_library_host_code_is_clean_under_asan_and_ubsan():
    p = subprocess.run([os.path.join(ROOT, "tools", "fakehip", "run.sh"), "--quick"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=1200)
    assert p.returncode == 0 and "sanitizer reports: 0" in p.stdout and "entry-point calls" in p.stdout, p.stdout[-3000:]
