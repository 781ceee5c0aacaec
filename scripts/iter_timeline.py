"""Dev tool: one training iteration out of a rocprofv3 kernel trace (csv): wall span between two adam_kernel launches, GPU-busy
time (union of the kernel intervals), idle time, the largest idle gaps with their neighbours, and kernel time by name.
usage: iter_timeline.py <kernel_trace.csv> [which_iteration_from_the_end=1] [min_gap_us=5]"""
import csv, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
thr = (float(sys.argv[3]) if len(sys.argv) > 3 else 5.0) * 1e3
adams = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
assert len(adams) >= which + 1, "need at least two adam_kernel launches"
lo, hi = adams[-which - 1] + 1, adams[-which]
it = rows[lo:hi + 1]
t0, t1 = rows[adams[-which - 1]][1], it[-1][1]
busy, end, gaps, last = 0, t0, [], "adam_kernel (previous iteration)"
for s, e, k in it:
    if s > end:
        gaps.append((s - end, end - t0, last, k))
        busy += e - s
    elif e > end:
        busy += e - end
    if e > end:
        end, last = e, k
print(f"iteration: {1e-6 * (t1 - t0):.3f} ms wall, {1e-6 * busy:.3f} ms busy, {1e-6 * (t1 - t0 - busy):.3f} ms idle in {len(gaps)} gaps, {len(it)} kernels")
big = sorted(gaps, reverse=True)
print(f"gaps >= {thr / 1e3:.0f} us: {sum(1 for g in gaps if g[0] >= thr)} totalling {1e-6 * sum(g[0] for g in gaps if g[0] >= thr):.3f} ms")
for g, at, a, b in big[:25]:
    print(f"  {1e-3 * g:8.1f} us at {1e-6 * at:7.2f} ms   after [{a[:60]}]  before [{b[:60]}]")
by = defaultdict(lambda: [0, 0])
for s, e, k in it:
    by[k][0] += e - s; by[k][1] += 1
print("kernel time by name:")
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"  {1e-6 * t:7.3f} ms {n:4d} x  {k[:110]}")
