#!/usr/bin/env python3
"""Per-kernel timeline of one bench step from a rocprofv3 --kernel-trace csv: start offset and duration of every
kernel of the LAST step (microseconds), gaps included -- shows what prepare() is made of."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_selfhist_card starts the last step
last = max(i for i, r in enumerate(rows) if "k_selfhist_card" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsh::", "")[:40]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-42s q%s  start %9.1f us  dur %9.1f us" % (name, r["Queue_Id"], s / 1e3, (e - s) / 1e3))
