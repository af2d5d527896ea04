#!/usr/bin/env python
"""
Instruction-issue summary of the PMC passes tools/gpu_round.sh collects (tools/pmc_pass.sh <tag>_lanes / <tag>_occ):

  profiles/<tag>_lanes.json   per kernel, mean per launch: SQ_INSTS_VALU / SALU / LDS, SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU,
                              SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES and lanes_active_per_valu_instruction =
                              SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
  profiles/pmc_issue.json     what bench.py reads for roofline.issue: VALU / SALU / LDS instructions per launch and active lanes per
                              VALU instruction of the dominant kernel (largest grid), with the grid size it was measured on

    python tools/lanes_summarise.py <tag> [dominant-kernel-substring] [workload]
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(path, acc, grids):
    if not os.path.isfile(path):
        return
    with open(path) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grids[k] = max(grids[k], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size"]) == grids[k]:                      # the big batch, not the single-scenario launches of the same kernel
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))


def main():
    tag = sys.argv[1]
    dominant = sys.argv[2] if len(sys.argv) > 2 else "k_paths<1"
    workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
    acc, grids = defaultdict(lambda: defaultdict(list)), defaultdict(int)
    for sub in ("lanes", "occ"):
        collect(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, sub), "p_counter_collection.csv"), acc, grids)
    out = {"what": "rocprofv3 --pmc passes over bench.py, mean per launch of the largest grid of every kernel; SQ_WAVE_CYCLES / "
                   "SQ_BUSY_CYCLES in units of 4 cycles", "tag": tag, "workload": workload, "kernels": {}}
    for k, d in acc.items():
        if "copyBuffer" in k or "fill" in k.lower():
            continue
        e = {c: sum(v) / len(v) for c, v in d.items()}
        if e.get("SQ_ACTIVE_INST_VALU"):
            e["lanes_active_per_valu_instruction"] = e.get("SQ_THREAD_CYCLES_VALU", 0.0) / e["SQ_ACTIVE_INST_VALU"]
        e["grid_size"] = grids[k]
        out["kernels"][k] = e
    dst = os.path.join(ROOT, "profiles")
    with open(os.path.join(dst, tag + "_lanes.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    dom = [k for k in out["kernels"] if dominant in k]
    if dom:
        e = out["kernels"][dom[0]]
        with open(os.path.join(dst, "pmc_issue.json"), "w") as fh:
            json.dump({"kernel": dom[0], "tag": tag, "workload": workload, "grid_size": e["grid_size"],
                       "valu_insts_per_launch": e.get("SQ_INSTS_VALU"), "salu_insts_per_launch": e.get("SQ_INSTS_SALU"),
                       "lds_insts_per_launch": e.get("SQ_INSTS_LDS"),
                       "valu_active_quad_cycles_per_launch": e.get("SQ_ACTIVE_INST_VALU"),    # quad-cycles a VALU instruction is in flight
                       "wave_quad_cycles_per_launch": e.get("SQ_WAVE_CYCLES"), "busy_quad_cycles_per_launch": e.get("SQ_BUSY_CYCLES"),
                       "lanes_active_per_valu_inst": e.get("lanes_active_per_valu_instruction"),
                       "waves_per_launch": e.get("SQ_WAVES")}, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
