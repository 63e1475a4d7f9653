#!/usr/bin/env python3
"""Per-kernel roofline table from committed profiles: rocprofv3 kernel stats (durations) joined with
the PMC traffic summary (HBM bytes per launch).  HBM-bound kernels are priced against 8 TB/s, the
MFMA convolutions are priced in bench.py (library-side flop counts)."""
import csv, json, re, sys

def main(stats_csv, traffic_json, steps, out):
    traffic = json.load(open(traffic_json))["kernels"]
    rows = []
    for r in csv.DictReader(open(stats_csv)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Name"]))
        t = traffic.get(name)
        avg_us = float(r["AverageNs"]) / 1e3
        ms_step = float(r["TotalDurationNs"]) / 1e6 / steps
        if t is None or ms_step < 0.15:
            continue
        gbs = t["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
        rows.append((ms_step, name, int(r["Calls"]) // steps, avg_us, t["hbm_bytes_per_launch"] / 1e6, gbs))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("| kernel | ms/step | launches/step | avg us | HBM MB/launch (PMC) | HBM GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
        for ms, name, n, us, mb, gbs in rows:
            f.write("| `%s` | %.2f | %d | %.1f | %.1f | %.0f | %.2f |\n" % (name[:70], ms, n, us, mb, gbs, gbs / 8000.0))
    print(open(out).read())

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
