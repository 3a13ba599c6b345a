#!/bin/bash
# HBM traffic of the klt / seeds / pose kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of
# `bench.py --workload <wl>` with its default sizes -> gpurun_out/profiles/<round>_<wl>_pmc_summary.json,
# which bench.py reads back (pmc_traffic, workload_key "<wl>:default").  Counters only: no trace flags beside --pmc.
set -e
ROUND=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles
mkdir -p $dst
for pair in klt:klt_track_kernel seeds:update_seeds_kernel pose:pose_optimize_kernel; do
  wl=${pair%%:*}; kern=${pair##*:}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=/tmp/pmc_${ROUND}_${wl}_$ctr
    rm -rf $out
    rocprofv3 --pmc $ctr --kernel-include-regex "$kern" --output-format csv -d $out -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
    cp $(find $out -name "*counter_collection.csv" | head -1) $dst/${ROUND}_${wl}_pmc_$ctr.csv
  done
  python - $dst $ROUND $wl $kern <<'PY'
import csv, sys, json
dst, rnd, wl, kern = sys.argv[1:5]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open("%s/%s_%s_pmc_%s.csv" % (dst, rnd, wl, ctr))) if r["Counter_Name"] == ctr and kern in r["Kernel_Name"]]
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    # the workload's own launches are the largest ones (bench.py also makes a few one-unit calls while warming up)
    top = [v for v in vals if v >= 0.5 * vals[-1]]
    res[ctr] = {"dispatches": len(top), "mean_per_dispatch_KB_as_reported": sum(top) / len(top)}
res["workload_key"] = "%s:default" % wl
res["kernel"] = kern
res["units"] = "FETCH_SIZE / WRITE_SIZE in KB (1024 B) per dispatch as rocprofv3 reports them; bench.py applies the gfx950 x2 to FETCH_SIZE"
json.dump(res, open("%s/%s_%s_pmc_summary.json" % (dst, rnd, wl), "w"), indent=1)
print(json.dumps(res))
PY
  rm -f $dst/${ROUND}_${wl}_pmc_FETCH_SIZE.csv $dst/${ROUND}_${wl}_pmc_WRITE_SIZE.csv
done
