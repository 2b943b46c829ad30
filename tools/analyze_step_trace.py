#!/usr/bin/env python
"""Timeline of the last complete training step in a rocprofv3 kernel trace (p_kernel_trace.csv): start offset,
duration, idle gap before each kernel, and the span / busy totals.  Usage: analyze_step_trace.py <csv> [steps_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", "")) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith("loss_kernel")]
step = ev[idx[-back - 1]:idx[-back]]
t0 = step[0][0]
busy_until, idle = step[0][0], 0
for s, e, n, q in step:
    gap = max(0, s - busy_until)
    idle += gap
    print("%8.1f %8.1f idle %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, n))
    busy_until = max(busy_until, e)
print("step span %.1f us, GPU idle %.1f us, kernels %d" % ((busy_until - t0) / 1e3, idle / 1e3, len(step)))
