"""What the device was doing during a traced run (scripts/profile_chain.py MODE=trace): for the steady part of the trace, the
share of the time at least one kernel / copy was running, the sum of the kernels' own durations against it (> 1: side by side),
the busy share of every hardware queue, and the mean duration of each kernel.  Usage: python scripts/device_timeline.py
<kernel_trace.csv> [<memory_copy_trace.csv>] [skip_fraction]"""
import csv, sys, collections

def load(path, name_col):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_col, ""), r.get("Queue_Id", r.get("Direction", ""))))
    return rows

def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot

k = load(sys.argv[1], "Kernel_Name")
m = load(sys.argv[2], "Name") if len(sys.argv) > 2 and sys.argv[2].endswith(".csv") else []
skip = float(sys.argv[-1]) if not sys.argv[-1].endswith(".csv") else 0.3
t0 = min(r[0] for r in k); t1 = max(r[1] for r in k)
lo = t0 + skip * (t1 - t0)
k = [r for r in k if r[0] >= lo]; m = [r for r in m if r[0] >= lo]
span = t1 - lo
ku = union([(r[0], r[1]) for r in k]); ks = sum(r[1] - r[0] for r in k)
print("steady span %.1f ms: a kernel running %.1f %% of it, kernel durations summed %.1f %% (side by side factor %.2f), %d dispatches (%.0f per ms)" %
      (span / 1e6, 100 * ku / span, 100 * ks / span, ks / max(1, ku), len(k), len(k) / (span / 1e6)))
if m:
    mu = union([(r[0], r[1]) for r in m]); ms = sum(r[1] - r[0] for r in m)
    au = union([(r[0], r[1]) for r in k + m])
    print("runtime copies: one in flight %.1f %% of the span (durations summed %.1f %%), %d copies; kernel or copy: %.1f %%" % (100 * mu / span, 100 * ms / span, len(m), 100 * au / span))
q = collections.defaultdict(list)
for r in k:
    q[r[3]].append((r[0], r[1]))
for qq, iv in sorted(q.items()):
    print("  queue %s: busy %.1f %%, %d dispatches" % (qq, 100 * union(iv) / span, len(iv)))
# gaps inside a chain: from the end of one dispatch of a queue to the start of its next one, where that is < 60 us (longer: the
# host was in between)
for qq, iv in sorted(q.items()):
    iv.sort()
    gaps = sorted(b[0] - a[1] for a, b in zip(iv, iv[1:]) if 0 <= b[0] - a[1] < 60000)
    if gaps:
        print("  queue %s: %d back-to-back gaps, median %.1f us, p90 %.1f us, sum %.1f %% of the span" %
              (qq, len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(0.9 * len(gaps))] / 1e3, 100 * sum(gaps) / span))
d = collections.defaultdict(list)
for r in k:
    d[r[2].split("(")[0][:90]].append(r[1] - r[0])
print("%-92s %8s %10s %10s" % ("kernel", "calls", "mean us", "total ms"))
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("%-92s %8d %10.1f %10.2f" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
