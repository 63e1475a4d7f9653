"""rocprofv3 --pmc passes of tools/pmc_bench.sh -> profiles/pmc_traffic.json: HBM bytes per launch and kernel
instance.  Units and corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE under-reports by an access-pattern dependent factor, so the factors are MEASURED in the same session
on kernels that touch a known byte count (tools/pmc_calib.cpp): the dword pixel gather of the convolution
kernels (calib_gather_b32), the 16-byte streaming read the guide quotes (calib_stream_b128, expected 2.0) and
the dword store of the epilogue (calib_write_b32).
    python tools/pmc_traffic.py gpurun_out/TAG profiles/pmc_traffic.json"""
import collections
import csv
import glob
import json
import re
import sys

KNOWN = 256 * (1 << 20) * 4


def load(dirglob, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(dirglob, recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            a = agg[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return agg


def csrc_hash():
    """sha256 over the convolution kernel sources (csrc/conv_igemm*, common.h) the counters were collected on (bench.py prints it in `traffic_source` and drops the
    traffic figure when the tree's hash differs: VERDICT r5 item 9)"""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "obj-gan_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.startswith("conv_igemm") or f == "common.h":           # the sources of the measured (convolution) kernels
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def main(prefix, out):
    cal_f = load(prefix + "_pmc_cal_fetch/**/*counter_collection.csv", "FETCH_SIZE")
    cal_w = load(prefix + "_pmc_cal_write/**/*counter_collection.csv", "WRITE_SIZE")

    def factor(agg, name):
        v, n = agg.get(name, [0.0, 0])
        return KNOWN / (1024.0 * v / n) if n and v else None
    f_stream, f_gather = factor(cal_f, "calib_stream_b128"), factor(cal_f, "calib_gather_b32")
    f_write = factor(cal_w, "calib_write_b32")
    fe = load(prefix + "_pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE")
    wr = load(prefix + "_pmc_write/**/*counter_collection.csv", "WRITE_SIZE")
    ff = f_gather or 2.0
    fw = f_write or 1.0
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        f, nf = fe.get(name, [0.0, 0])
        w, nw = wr.get(name, [0.0, 0])
        if not nf and not nw:
            continue
        fb = ff * 1024.0 * f / max(nf, 1)
        wb = fw * 1024.0 * w / max(nw, 1)
        kernels[name] = {"launches": max(nf, nw), "fetch_bytes_per_launch": round(fb),
                         "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb)}
    json.dump({"conv_math": "fp16x2", "per_gpu_batch": 16,      # the bench defaults tools/pmc_bench.sh runs
               "csrc_sha16": csrc_hash(),
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py "
                         "(tools/pmc_bench.sh)",
               "calibration": {"known_bytes": KNOWN, "fetch_factor_stream_b128": f_stream,
                               "fetch_factor_gather_b32": f_gather, "write_factor_b32": f_write,
                               "applied": "KiB -> bytes; FETCH x %.3f (measured dword-gather factor), WRITE x %.3f"
                                          % (ff, fw)},
               "kernels": kernels}, open(out, "w"), indent=1)
    print("calibration: stream b128 x%s  gather b32 x%s  write b32 x%s" % (f_stream, f_gather, f_write))
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print("%-48s x%5d  %8.1f MB/launch" % (k[:48], v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
