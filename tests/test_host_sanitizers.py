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


def test_allocation_failures_come_back_as_status_codes():
    """include/ltpl_hip.h: "No C++ exception crosses the ABI". With the address space capped (RLIMIT_AS) the host containers of the entry
    points fail to allocate: std::bad_alloc must surface as LTPL_ERR_EXCEPTION + message, not as an abort (tools/fakehip/alloc_failure.py,
    library host code on the stand-in runtime, no sanitizers: they do not get along with an address-space limit)."""
    import sys
    env = dict(os.environ, FAKEHIP_SAN="none", LTPL_NO_SELFTEST="1")
    env.pop("LD_PRELOAD", None)                    # (when the suite itself runs under tools/sanitize_host.sh)
    subprocess.run([os.path.join(ROOT, "tools", "fakehip", "build.sh")], check=True, env=env, stdout=subprocess.DEVNULL, timeout=900)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fakehip", "alloc_failure.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert p.returncode == 0 and "alloc-failure check OK" in p.stdout and "C++ exception caught at the ABI" in p.stdout, p.stdout[-3000:]
