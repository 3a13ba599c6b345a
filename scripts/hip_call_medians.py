"""Median and total duration of every HIP runtime call of a rocprofv3 --hip-trace run, per call name, plus calls per
second over the traced interval: how busy the runtime's call path is (threads that enter it at the same time queue up).
  python scripts/hip_call_medians.py <hip_api_trace.csv> [skip_first_ms]"""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
b = [int(r["Start_Timestamp"]) for r in rows]
t_min, t_max = min(b), max(int(r["End_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
tids = set()
n = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t_min) * 1e-6 < skip:
        continue
    by[r["Function"]].append((e - s) * 1e-3)
    tids.add(r["Thread_Id"])
    n += 1
span_ms = (t_max - t_min) * 1e-6 - skip
print("%d calls by %d threads in %.1f ms = %.0f calls/s; time inside calls (all threads): %.1f ms" % (n, len(tids), span_ms, n / span_ms * 1e3, sum(sum(v) for v in by.values()) * 1e-3))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        print("  %-28s %6d calls  median %8.1f us  p90 %8.1f us  total %8.2f ms" % (k, len(v), np.median(v), np.percentile(v, 90), sum(v) * 1e-3))
