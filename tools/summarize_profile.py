#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>, written by tools/profile_round.sh) into the small files
committed under profiles/.

usage: python tools/summarize_profile.py gpurun_out/r03 profiles/r03        (workloads: bench, msa; round 5: pbatch, pbatchfetch)
Writes <prefix>_<workload>_kernel_stats.csv (copies of the --stats summaries) and <prefix>_pmc_summary.json:
per workload and kernel the sums of every collected counter, per-launch HBM traffic with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE x2 for wide coalesced reads, both counters in KiB), VALU instructions per wave, and
the launch shape (grid) the figures belong to.  The top-level keys nw_fill_bits ... are the bench workload's kernels
(what bench.py reads for roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = ["nw_fill_bits", "nw_traceback_windows", "nw_fill_cells", "nw_tb_scout", "nw_tb_resolve", "nw_tb_emit", "nw_tb_gather", "nw_pack_planes", "nw_expand_rows", "sp_columns"]


def kernel_key(name):
    for k in KERNELS:
        if k in name:
            return k
    return "other"


def find(src, stem, suffix):
    hits = glob.glob(os.path.join(src, "**", stem + "_" + suffix), recursive=True) + glob.glob(os.path.join(src, stem + "_" + suffix))
    return hits[0] if hits else None


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    out = {}
    for wl in ("bench", "msa", "pbatch", "pbatchfetch"):
        for variant in ("stats", "solo"):
            st = find(src, "%s_%s" % (wl, variant), "kernel_stats.csv")
            if st:
                shutil.copy(st, "%s_%s_kernel_%s.csv" % (prefix, wl, variant))
        res = {}
        for fn in sorted(glob.glob(os.path.join(src, "**", "%s_pmc_*_counter_collection.csv" % wl), recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            disp = collections.defaultdict(set)
            dur = collections.defaultdict(dict)
            grid = collections.defaultdict(collections.Counter)
            for r in csv.DictReader(open(fn)):
                key = kernel_key(r["Kernel_Name"])
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[key].add(r["Dispatch_Id"])
                dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                grid[key][(r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))] += 1
            for key in agg:
                o = res.setdefault(key, {"counters": {}})
                o["dispatches"] = len(disp[key])
                o["avg_dispatch_us_under_pmc"] = round(sum(dur[key].values()) / max(len(dur[key]), 1) / 1e3, 2)
                o["grids"] = {"%s x %s" % g: n for g, n in grid[key].most_common(4)}
                for c, v in agg[key].items():
                    o["counters"][c] = v
        for key, o in res.items():
            c = o["counters"]
            n = max(o["dispatches"], 1)
            if "WRITE_SIZE" in c:
                o["hbm_write_bytes_per_launch"] = round(c["WRITE_SIZE"] * 1024 / n)
            if "FETCH_SIZE" in c:
                o["hbm_read_bytes_per_launch_x2_corrected"] = round(2 * c["FETCH_SIZE"] * 1024 / n)
            if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
                o["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
                o["valu_insts_per_launch"] = round(c["SQ_INSTS_VALU"] / n)
            if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
                o["wave_cycles_per_wave_x4"] = round(4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"])
            if "SQ_LDS_BANK_CONFLICT" in c and "SQ_INSTS_LDS" in c and c["SQ_INSTS_LDS"]:
                o["lds_conflict_cycles_per_lds_inst"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"], 3)
        if res:
            out[wl] = res
    if "bench" in out:                      # what bench.py reads: the bench workload's kernels at top level
        for key, o in out["bench"].items():
            top = dict(o)
            if key == "nw_fill_bits":       # merged passes of 128 pairs of 16384 letters: jobs = workgroups of the launch
                grid = next(iter(o.get("grids", {"0 x 1": 1})))
                g, w = grid.split(" x ")
                top["launch_shape"] = {"jobs": int(g) // max(int(w), 1), "len": 16384}
            out[key] = top
    with open(prefix + "_pmc_summary.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in out.items() if k in ("nw_fill_bits", "nw_traceback_windows")}, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
