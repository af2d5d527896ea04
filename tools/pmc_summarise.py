#!/usr/bin/env python
"""One rocprofv3 --pmc pass of tools/pmc_ab.sh -> per-scenario figures on stdout and the per-launch means accumulated into

  gpurun_out/pmc_issue_<variant>[_<workload>].json     SQ instruction / cycle counters  (bench.py: profiles/pmc_issue[_c3].json)
  gpurun_out/pmc_traffic_<variant>[_<workload>].json   FETCH_SIZE / WRITE_SIZE          (bench.py: profiles/pmc_traffic[_c3].json)

Every file carries `build` = __graft_entry__.build_stamp of the PROFILED library (digest of the batch path kernel's instruction stream,
its register metadata, sha256 of the file): bench.py recomputes the digest for the library it runs and reports `build_matches`.
HBM bytes: FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts
128-B requests as 64 B for wide coalesced reads -> x 2 is the upper bound that is reported as `hbm_bytes_per_launch`, WRITE_SIZE as is.

    python tools/pmc_summarise.py <p_counter_collection.csv> <variant> <workload> <library>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge                                                   # noqa: E402

ISSUE_KEYS = {"SQ_INSTS_VALU": "valu_insts_per_launch", "SQ_INSTS_SALU": "salu_insts_per_launch", "SQ_INSTS_LDS": "lds_insts_per_launch",
              "SQ_INSTS_SMEM": "smem_insts_per_launch", "SQ_INSTS_VMEM_RD": "vmem_rd_insts_per_launch", "SQ_INSTS_VMEM_WR": "vmem_wr_insts_per_launch",
              "SQ_ACTIVE_INST_VALU": "valu_active_quad_cycles_per_launch",
              "SQ_ACTIVE_INST_LDS": "lds_active_quad_cycles_per_launch", "SQ_ACTIVE_INST_SCA": "scalar_active_quad_cycles_per_launch",
              "SQ_ACTIVE_INST_ANY": "any_active_quad_cycles_per_launch",
              "SQ_WAVE_CYCLES": "wave_quad_cycles_per_launch", "SQ_WAIT_ANY": "wait_any_quad_cycles_per_launch",
              "SQ_WAIT_INST_ANY": "wait_inst_quad_cycles_per_launch", "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict_cycles_per_launch",
              "SQ_BUSY_CYCLES": "sq_busy_quad_cycles_per_launch"}
MANGLED = {"PlanRtG": "_Z7k_pathsILi1E7PlanRtGE", "PlanRt": "_Z7k_pathsILi1E6PlanRtE"}


def mangled_of(kernel_name):
    """'k_paths<1, PlanFx<32, 32, 1> >(...)' -> the symbol prefix ltpl_paths_kernel_symbol reports for it"""
    import re
    m = re.search(r"PlanFx<\s*(\d+),\s*(\d+),\s*(\d+)\s*>", kernel_name)
    if m:
        return "_Z7k_pathsILi1E6PlanFxILi%sELi%sELi%sEEE" % m.groups()
    return MANGLED["PlanRtG"] if "PlanRtG" in kernel_name else MANGLED["PlanRt"]


def main():
    path, variant, workload, lib = sys.argv[1:5]
    rows = [r for r in csv.DictReader(open(path)) if "k_paths<1" in r["Kernel_Name"]]
    if not rows:
        print(variant, "no k_paths<1 dispatch in", path)
        return 1
    grid = max(int(r["Grid_Size"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size"]) == grid]            # the full-batch launches (ltpl_create's self-test launches 64 scenarios)
    n_scen = grid // 64
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    per = {c: sum(v) / len(v) for c, v in acc.items()}
    print(variant, workload, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v / n_scen) for c, v in sorted(per.items())), "(per scenario, %d launches)" % (len(rows) // max(len(acc), 1)))
    suffix = "" if workload == "c2" else "_" + workload
    stamp = ge.build_stamp(os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib, mangled_of(rows[0]["Kernel_Name"]))
    tag = os.environ.get("PMC_TAG", "r05")
    base = {"kernel": rows[0]["Kernel_Name"].split("(")[0].replace("void ", ""), "tag": tag, "workload": workload, "grid_size": grid,
            "library": lib, "build": stamp}

    def update(name, what, fields):
        dst = os.path.join(ROOT, "gpurun_out", "%s_%s%s.json" % (name, variant, suffix))
        d = json.load(open(dst)) if os.path.isfile(dst) else {}
        if (d.get("build") or {}).get("isa_sha256") != stamp["isa_sha256"] or d.get("grid_size") != grid:
            d = {}                                                             # a file of another build / batch: start over
        d.update(base); d["what"] = what; d.update(fields)
        json.dump(d, open(dst, "w"), indent=1)

    issue = {k: per[c] for c, k in ISSUE_KEYS.items() if c in per}
    if "SQ_THREAD_CYCLES_VALU" in per and per.get("SQ_ACTIVE_INST_VALU"):
        issue["lanes_active_per_valu_inst"] = per["SQ_THREAD_CYCLES_VALU"] / per["SQ_ACTIVE_INST_VALU"]
    extra = {c: per[c] for c in per if c not in ISSUE_KEYS and c not in ("SQ_THREAD_CYCLES_VALU", "FETCH_SIZE", "WRITE_SIZE")}
    if extra:
        issue["other_counters_per_launch"] = extra
    if issue:
        update("pmc_issue", "rocprofv3 --pmc passes over tools/dbg_plain_paths.py (path kernel alone), mean per launch; SQ_*_CYCLES / "
                            "SQ_ACTIVE_* in units of 4 cycles (quad-cycles)", issue)
    tr = {}
    if "FETCH_SIZE" in per:
        tr["fetch_bytes_raw"] = per["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in per:
        tr["write_bytes_raw"] = per["WRITE_SIZE"] * 1024.0
    if tr:
        dst = os.path.join(ROOT, "gpurun_out", "pmc_traffic_%s%s.json" % (variant, suffix))
        old = json.load(open(dst)) if os.path.isfile(dst) else {}
        if (old.get("build") or {}).get("isa_sha256") == stamp["isa_sha256"] and old.get("grid_size") == grid:
            for k in ("fetch_bytes_raw", "write_bytes_raw"):
                if k in old and k not in tr:
                    tr[k] = old[k]
        f, w = tr.get("fetch_bytes_raw"), tr.get("write_bytes_raw")
        if f is not None and w is not None:
            tr["hbm_bytes_per_launch_raw"] = f + w
            tr["hbm_bytes_per_launch"] = 2.0 * f + w
        update("pmc_traffic", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per dispatch) over tools/dbg_plain_paths.py, mean per "
                              "launch; hbm_bytes_per_launch = 2 x fetch + write (gfx950: FETCH_SIZE under-reports wide coalesced reads by up to "
                              "2 x, MI355X_MICROARCH.md; WRITE_SIZE as is)", tr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
