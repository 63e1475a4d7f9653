#!/usr/bin/env python3
"""Whole-step roofline table from the three PMC passes of tools/pmc_step.sh (aggregated on the GPU box by
tools/pmc_step.py) and a rocprofv3 --stats summary of the same bench command: per kernel the time per step,
HBM bytes per launch (FETCH_SIZE / WRITE_SIZE in KiB; FETCH x2.0, WRITE x1.0 -- the factors measured in
profiles/r02_pmc_calibration_and_traffic.txt), achieved HBM GB/s against 8 TB/s, and the MFMA-pipe occupancy
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8) of the same pass (GRBM_GUI_ACTIVE is
reported once per XCD and summed over the 8 of them; cross-checked in round 1 against time x clock,
profiles/r01_tm1_l2_bound.txt).
    python tools/roofline_step.py gpurun_out/TAG_pmcstep.json profiles/STATS.csv STEPS out.md"""
import csv
import json
import re
import sys

FETCH_FACTOR, WRITE_FACTOR, SIMDS, XCDS = 2.0, 1.0, 4 * 256, 8


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def main(pmc_json, stats_csv, steps, out):
    d = json.load(open(pmc_json))
    steps = int(steps)
    rows = []
    for r in csv.DictReader(open(stats_csv)):
        k = short(r["Name"])
        ms_step = float(r["TotalDurationNs"]) / 1e6 / steps
        if ms_step < 0.15:
            continue
        avg_us = float(r["AverageNs"]) / 1e3
        f, w, m = d["pass1"].get(k), d["pass2"].get(k), d["pass3"].get(k)
        fb = FETCH_FACTOR * 1024.0 * f["FETCH_SIZE"] / f["launches"] if f and f.get("launches") else None
        wb = WRITE_FACTOR * 1024.0 * w["WRITE_SIZE"] / w["launches"] if w and w.get("launches") else None
        hbm = (fb or 0.0) + (wb or 0.0) if (fb is not None or wb is not None) else None
        gbs = hbm / (avg_us * 1e-6) / 1e9 if hbm else None
        util = None
        if m and m.get("GRBM_GUI_ACTIVE"):
            util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * m["GRBM_GUI_ACTIVE"] / XCDS)
        rows.append((ms_step, k, int(r["Calls"]) / steps, avg_us, hbm, gbs, util))
    rows.sort(reverse=True)
    with open(out, "w") as fo:
        fo.write("| kernel | ms/step | launches/step | avg us | HBM MB/launch | HBM GB/s | of 8 TB/s | MFMA pipe busy |\n"
                 "|---|---|---|---|---|---|---|---|\n")
        for ms, k, n, us, hbm, gbs, util in rows:
            fo.write("| `%s` | %.2f | %.1f | %.1f | %s | %s | %s | %s |\n" % (
                k[:64], ms, n, us, "%.1f" % (hbm / 1e6) if hbm else "-", "%.0f" % gbs if gbs else "-",
                "%.2f" % (gbs / 8000.0) if gbs else "-", "%.0f %%" % (100 * util) if util else "-"))
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:5])
