#!/usr/bin/env python3
"""Occupancy timeline of a bench run from a rocprofv3 --kernel-trace CSV: how much of the wall time of the last steps has
(a) no kernel in flight, (b) only small kernels (< BIG us) in flight, (c) at least one big kernel in flight -- and the sum of
kernel durations by class over the same window.  usage: timeline.py <dir> [steps_in_run] [big_us]"""
import csv, glob, re, sys

def main():
    d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4; big = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:50]))
    rows.sort()
    # window: `steps` x 150 ms ending 30 ms before the last kernel of the run (inside the timed steps of bench.py)
    t_hi = rows[-1][1] - 30e6; t_lo = t_hi - steps * 150e6
    rows = [r for r in rows if r[0] < t_hi]
    win = [(max(s, t_lo), min(e, t_hi), n) for s, e, n in rows if e > t_lo]
    ev = []
    for s, e, n in win:
        b = (e - s) / 1e3 >= big
        ev.append((s, 1, b)); ev.append((e, -1, b))
    ev.sort()
    nb = ns = 0; last = t_lo; idle = small = bigt = 0.0
    for t, dlt, b in ev:
        dt = t - last
        if nb > 0: bigt += dt
        elif ns > 0: small += dt
        else: idle += dt
        last = t
        if b: nb += dlt
        else: ns += dlt
    # who fills the time without a big kernel: sweep again, charging every such interval to the small kernels in flight
    # (split evenly) and every idle gap to the kernel that starts next
    from collections import defaultdict
    charge = defaultdict(float); gapnext = defaultdict(float)
    ev2 = []
    for i, (s, e, n) in enumerate(win):
        b = (e - s) / 1e3 >= big
        ev2.append((s, 1, b, n)); ev2.append((e, -1, b, n))
    ev2.sort(key=lambda x: (x[0], x[1]))
    nb = 0; live = defaultdict(int); last = t_lo; pending_gap = 0.0
    for t, dlt, b, n in ev2:
        dt = t - last
        if nb == 0 and dt > 0:
            tot = sum(live.values())
            if tot > 0:
                for k, c in live.items():
                    if c > 0: charge[k] += dt * c / tot
            else:
                pending_gap += dt
        last = t
        if dlt > 0 and pending_gap > 0:
            gapnext[n] += pending_gap; pending_gap = 0.0
        if b: nb += dlt
        else:
            live[n] += dlt
    wall = (t_hi - t_lo) / 1e6
    ksum_big = sum(e - s for s, e, n in win if (e - s) / 1e3 >= big) / 1e6
    ksum_small = sum(e - s for s, e, n in win if (e - s) / 1e3 < big) / 1e6
    nsmall = sum(1 for s, e, n in win if (e - s) / 1e3 < big)
    print("window %.1f ms: idle %.1f ms (%.1f%%), only kernels < %.0f us in flight %.1f ms (%.1f%%), >= 1 big kernel in flight %.1f ms (%.1f%%)"
          % (wall, idle / 1e6, 100 * idle / 1e6 / wall, big, small / 1e6, 100 * small / 1e6 / wall, bigt / 1e6, 100 * bigt / 1e6 / wall))
    print("kernel-duration sums in the window: big %.1f ms, small %.1f ms (%d launches)" % (ksum_big, ksum_small, nsmall))
    print("-- time without a big kernel in flight, charged to the small kernels running (ms per window):")
    for k, v in sorted(charge.items(), key=lambda kv: -kv[1])[:28]:
        print("   %8.2f  %s" % (v / 1e6, k))
    print("-- idle gaps, charged to the kernel that starts next (ms per window):")
    for k, v in sorted(gapnext.items(), key=lambda kv: -kv[1])[:16]:
        print("   %8.2f  %s" % (v / 1e6, k))

if __name__ == "__main__":
    main()
