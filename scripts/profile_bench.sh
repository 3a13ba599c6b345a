#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace/stats of bench.py, then two PMC
# passes (FETCH_SIZE, WRITE_SIZE) of the same command; small summaries are left
# under gpurun_out/profiles/ (the raw traces stay in /tmp).
set -e
ROUND=${1:-r01}
ARGS=${2:---steps 5 --warmup 2 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles
mkdir -p $dst
out=/tmp/prof_$ROUND
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py $ARGS > $out.trace.log 2>&1 || { tail -20 $out.trace.log; exit 1; }
grep '^{' $out.trace.log > $dst/${ROUND}_bench_under_rocprof.json || true
st=$(find $out/trace -name "*kernel_stats.csv" | head -1)
tr=$(find $out/trace -name "*kernel_trace.csv" | head -1)
(head -1 $st; grep -E "svoh::" $st) > $dst/${ROUND}_kernel_stats_svoh.csv
head -12 $st > $dst/${ROUND}_kernel_stats_top.csv
(head -1 $tr; grep -E "svoh::" $tr) > $dst/${ROUND}_kernel_trace_svoh.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-include-regex "sparse_align" --output-format csv -d $out/$ctr -- python bench.py $ARGS > $out.$ctr.log 2>&1 || { tail -20 $out.$ctr.log; exit 1; }
  f=$(find $out/$ctr -name "*counter_collection.csv" | head -1)
  cp $f $dst/${ROUND}_pmc_$ctr.csv
done
python - $dst $ROUND <<'PY'
import csv, sys, json
dst, rnd = sys.argv[1], sys.argv[2]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    # the batch kernel only: bench.py also times one frame pair alone (the cluster-mode instantiation, "..., true>")
    rows = [r for r in csv.DictReader(open("%s/%s_pmc_%s.csv" % (dst, rnd, ctr)))
            if r["Counter_Name"] == ctr and ", true>(" not in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    res[ctr] = {"dispatches": len(vals), "mean_per_dispatch_KB_as_reported": sum(vals) / len(vals)}
st = list(csv.DictReader(open("%s/%s_kernel_stats_svoh.csv" % (dst, rnd))))
res["kernel_stats"] = st
res["workload_key"] = "align:B4096:N2000:P4:L4-0"  # the default bench.py workload these passes ran (4096 frame pairs per step since the end of round 6)
res["units"] = "FETCH_SIZE / WRITE_SIZE in KB (1024 B) per dispatch as rocprofv3 reports them; bench.py applies the gfx950 x2 to FETCH_SIZE"
json.dump(res, open("%s/%s_summary.json" % (dst, rnd), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
