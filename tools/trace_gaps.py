#!/usr/bin/env python3
"""Kernel start / end times of the LAST step in a rocprofv3 kernel trace (csv) of tools/rank_trace.py: every kernel with
its duration and the idle time in front of it.  usage: trace_gaps.py <kernel_trace.csv> [kernels per step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]) for x in rows), key=lambda t: t[0])
# steps are separated by the largest gaps (host synchronisation): take the kernels behind the last such gap
gaps = [(ks[i][0] - max(k[1] for k in ks[:i]), i) for i in range(1, len(ks))]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if per:
    last = ks[-per:]
else:
    big = sorted(gaps, reverse=True)[: max(1, len(ks) // 8)]
    cut = max(i for g, i in big if g > 20000) if any(g > 20000 for g, _ in big) else 0
    last = ks[cut:]
t0 = last[0][0]
end = t0
busy = 0
for s, e, name in last:
    short = name.split("(")[0].replace("void dsh::", "")[:48]
    print("%9.1f us  +%7.1f idle  %8.1f us  %s" % ((s - t0) / 1e3, max(0, s - end) / 1e3, (e - s) / 1e3, short))
    busy += e - s
    end = max(end, e)
print("step %.1f us, kernels %.1f us, %d launches" % ((end - t0) / 1e3, busy / 1e3, len(last)))
