import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Function"] in ("hipFree","hipMalloc","hipHostMalloc","hipHostFree","hipStreamCreateWithFlags")]
t0=min(int(r["Start_Timestamp"]) for r in rows)
b=collections.defaultdict(lambda: collections.Counter())
for r in rows:
    b[(int(r["Start_Timestamp"])-t0)//100_000_000][r["Function"]]+=1
for k in sorted(b): print("%5.1f s" % (k/10), dict(b[k]))
