"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV.
python tools/gap_analysis.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
print(f"{len(rows)} kernels in {path}")
busy = sum(e - s for s, e, _ in rows)
gaps = []
by_prev = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 200_000:          # larger: host-side pauses between steps / phases, not launch gaps
        gaps.append(g)
        by_prev[n0].append(g)
gaps.sort()
n = len(gaps)
print(f"busy {busy / 1e6:.1f} ms; gaps<200us: n={n} total {sum(gaps) / 1e6:.2f} ms ({100.0 * sum(gaps) / (busy + sum(gaps)):.1f} % of busy+gaps); "
      f"median {gaps[n // 2] / 1e3:.2f} us, p90 {gaps[int(n * 0.9)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us; negative (overlap): {sum(1 for g in gaps if g < 0)}")
for name, g in sorted(by_prev.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"  after {name:60s} n={len(g):5d} mean {sum(g) / len(g) / 1e3:6.2f} us  total {sum(g) / 1e6:6.2f} ms")
