#!/usr/bin/env python
"""GPU-side accounting of a rocprofv3 --kernel-trace CSV of `bench.py`: busy time, idle gaps (> 20 us) and what runs
right after each gap.  usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys, collections

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, n in rows:
    if s > cur_end:
        if (s - cur_end) / 1e3 >= min_gap:
            gaps.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e6, n))
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms; GPU busy {busy / 1e6:.1f} ms; {len(gaps)} gaps >= {min_gap} us totalling "
      f"{sum(g[0] for g in gaps) / 1e3:.1f} ms")
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
agg = collections.Counter()
for g, at, n in gaps:
    agg[short(n)] += g
print("idle time by the kernel that ends the gap (ms):")
for n, g in agg.most_common(25):
    print(f"  {g / 1e3:8.2f}  {n}")
big = sorted(gaps, reverse=True)[:15]
print("largest gaps: us, at ms, next kernel")
for g, at, n in big:
    print(f"  {g:9.1f}  {at:9.1f}  {short(n)}")
