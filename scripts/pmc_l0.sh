#!/bin/bash
# L2-side counters of the level-0 pass of the alignment kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for CTR in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "FETCH_SIZE"; do
  T=$(echo $CTR | cut -d' ' -f1)
  rm -rf /tmp/pmc_$T
  (cd $R && rocprofv3 --pmc $CTR --kernel-include-regex "sparse_align_kernel" --output-format csv -d /tmp/pmc_$T -- python scripts/perf_l0.py > /tmp/pmc_$T.log 2>&1 || tail -3 /tmp/pmc_$T.log)
  tail -1 /tmp/pmc_$T.log
  F=$(find /tmp/pmc_$T -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv,sys,collections
try:
    rows=list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
agg=collections.defaultdict(list)
for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print("  %-32s mean per dispatch %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
done
