#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>) into the small files committed under profiles/.

usage: python tools/summarize_profile.py gpurun_out/r01 profiles/r01
Writes <prefix>_kernel_stats.csv (copy of the --stats summary), <prefix>_pmc_summary.json
(per-kernel sums of every collected counter, per-launch HBM traffic with the gfx950
corrections of MI355X_MICROARCH.md: FETCH_SIZE x2 for wide coalesced reads, both in KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    stats = os.path.join(src, "stats_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, prefix + "_kernel_stats.csv")
    out = {}
    for fn in sorted(glob.glob(os.path.join(src, "pmc_*_counter_collection.csv"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"]
            key = ("nw_fill_bits" if "nw_fill_bits" in name else "nw_traceback_replay" if "nw_traceback_replay" in name else "nw_traceback_bits" if "nw_traceback_bits" in name else "nw_fill_tiles_pk" if "nw_fill_tiles_pk" in name else "nw_fill_tiles" if "nw_fill_tiles" in name else
                   "nw_traceback_pk" if "nw_traceback_pk" in name else "nw_traceback" if "nw_traceback" in name else "other")
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[key].add(r["Dispatch_Id"])
            dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for key in agg:
            o = out.setdefault(key, {"counters": {}})
            o["dispatches"] = len(disp[key])
            o["avg_dispatch_us_under_pmc"] = round(sum(dur[key].values()) / max(len(dur[key]), 1) / 1e3, 2)
            for c, v in agg[key].items():
                o["counters"][c] = v
    for key, o in out.items():
        c = o["counters"]
        n = max(o["dispatches"], 1)
        if "WRITE_SIZE" in c:
            o["hbm_write_bytes_per_launch"] = round(c["WRITE_SIZE"] * 1024 / n)
        if "FETCH_SIZE" in c:
            o["hbm_read_bytes_per_launch_x2_corrected"] = round(2 * c["FETCH_SIZE"] * 1024 / n)
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            o["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
            o["wave_cycles_per_wave_x4"] = round(4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"])
        for num in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if num in c and "SQ_WAVE_CYCLES" in c:
                pass
    with open(prefix + "_pmc_summary.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
