#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid) -- one row per launch shape.

usage: trace_agg.py <dir-with-*_kernel_trace.csv> [top]
"""
import csv, glob, re, sys
from collections import defaultdict

def main():
    d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name)
            grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            k = (name[:60], grid)
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel ms {tot:.2f} over {sum(v[0] for v in agg.values())} launches")
    for (name, grid), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{ms:9.3f} ms {n:5d}x {ms / n * 1e3:9.1f} us  {name}  grid={grid}")

if __name__ == "__main__":
    main()
