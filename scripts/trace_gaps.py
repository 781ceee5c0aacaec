"""Dev tool: idle gaps of the GPU in a rocprofv3 kernel trace (csv): for every gap above a threshold, the kernel that ended
before it and the one that started after it.   usage: trace_gaps.py <kernel_trace.csv> [min_gap_ms]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
end, last = rows[0][1], rows[0][2]
t0 = rows[0][0]
n = 0
for s, e, k in rows[1:]:
    if s - end > thr:
        n += 1
        print(f"gap {1e-6 * (s - end):8.2f} ms at {1e-6 * (end - t0):9.2f} ms   after [{last}]   before [{k}]")
    if e > end:
        end, last = e, k
print(n, "gaps above", thr / 1e6, "ms; trace spans", 1e-6 * (end - t0), "ms,", len(rows), "kernels")
