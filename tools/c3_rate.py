"""C3 throughput of the resident tick pipeline (bench.c3_throughput) for one or more batch sizes: tools/c3_rate.py [batch ...]
(LTPL_HIP_LIB selects a library variant; parity is checked on 64 scenarios per run)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [8192]:
    r = bench.c3_throughput(n, n_parity=64)
    print("c3 batch %6d  %.2f M ticks/s  ms/step %.4f  k_paths live %.4f alone %.4f  frac %.3f  parity %s  max_rel %.2e  lib %s" % (
        n, r["ticks_per_s"] / 1e6, r["ms_per_step"], r["kernel_ms"], r["kernel_ms_not_overlapped"], r["roofline_frac"], r.get("parity_checked"),
        r["parity_detail"]["max_rel_err"], os.environ.get("LTPL_HIP_LIB", "base")))
