#!/usr/bin/env python3
"""Print the dispatch timeline of one frame from a rocprofv3 kernel_trace.csv (start offsets, durations, gaps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = len(rows) // 2
while "k_gate_bilateral" not in rows[i]["Kernel_Name"]:
    i += 1
t0 = int(rows[i]["Start_Timestamp"]); prev_end = t0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
for r in rows[i:i + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s wg=%-6s grid=%-8s start=%8.1f dur=%7.1f gap=%6.1f us" % (r["Kernel_Name"][:30], r.get("Workgroup_Size", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
