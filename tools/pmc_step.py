"""Aggregate the three passes of tools/pmc_step.sh per kernel name (run on the GPU box; the per-dispatch
tables stay there): launches, summed counter values and summed durations of each pass.
    python tools/pmc_step.py /tmp/TAG_pmcstep out.json"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def main(prefix, out):
    res = {}
    for i in (1, 2, 3):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        for f in glob.glob("%s%d/**/*counter_collection.csv" % (prefix, i), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                seen[k].add(row.get("Dispatch_Id") or len(seen[k]))
        dur = collections.defaultdict(float)
        for f in glob.glob("%s%d/**/*kernel_trace.csv" % (prefix, i), recursive=True):
            for row in csv.DictReader(open(f)):
                dur[short(row["Kernel_Name"])] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        res["pass%d" % i] = {k: dict(v, launches=len(seen[k]), total_us=round(dur.get(k, 0.0), 1))
                             for k, v in agg.items()}
    json.dump(res, open(out, "w"), indent=0)
    print("kernels per pass:", [len(v) for v in res.values()])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
