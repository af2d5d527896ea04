#!/usr/bin/env python
"""
Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the committed summaries:

  profiles/<tag>_kernel_stats.csv   verbatim `rocprofv3 --kernel-trace --stats` per-kernel table of the bench command
  profiles/<tag>_pmc.json           per-kernel HBM traffic from the two PMC passes (FETCH_SIZE, WRITE_SIZE; KiB units),
                                    raw and with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM
                                    (FETCH_SIZE under-reports wide coalesced reads by 2x; WRITE_SIZE uncalibrated)
  profiles/pmc_traffic.json         what bench.py reads for roofline.traffic: corrected HBM bytes per launch of the
                                    dominant kernel

    python tools/profile_summarise.py <tag> [dominant-kernel-substring]
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel_counter(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            acc[row["Kernel_Name"]].append((float(row["Counter_Value"]), int(row["Grid_Size"])))
    return acc


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    dominant = sys.argv[2] if len(sys.argv) > 2 else "k_paths<1"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(os.path.join(src, "stats", "k_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    fetch = per_kernel_counter(os.path.join(src, "fetch", "k_counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel_counter(os.path.join(src, "write", "k_counter_collection.csv"), "WRITE_SIZE")
    out = {"tag": tag, "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch",
           "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 upper bound; "
                         "WRITE_SIZE uncalibrated (taken as is)",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f = [v for v, _ in fetch.get(name, [])]
        w = [v for v, _ in write.get(name, [])]
        grid = max([g for _, g in fetch.get(name, [])] + [g for _, g in write.get(name, [])])
        # the bench launches the big batch and single-scenario ticks through different kernels; keep the largest grid
        fb = [v for v, g in fetch.get(name, []) if g == grid]
        wb = [v for v, g in write.get(name, []) if g == grid]
        fm = sum(fb) / len(fb) * 1024 if fb else None
        wm = sum(wb) / len(wb) * 1024 if wb else None
        out["kernels"][short(name)] = {
            "grid_size": grid, "dispatches_fetch_pass": len(f), "dispatches_write_pass": len(w),
            "fetch_bytes_raw": fm, "write_bytes_raw": wm,
            "hbm_bytes_raw": (fm or 0) + (wm or 0),
            "hbm_bytes_corrected": 2 * (fm or 0) + (wm or 0),
        }
    with open(os.path.join(dst, tag + "_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    dom = [k for k in out["kernels"] if dominant in k]
    if dom:
        k = out["kernels"][dom[0]]
        with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
            json.dump({"kernel": dom[0], "tag": tag, "hbm_bytes_per_launch": k["hbm_bytes_corrected"],
                       "hbm_bytes_per_launch_raw": k["hbm_bytes_raw"], "grid_size": k["grid_size"]}, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
