#!/bin/bash
# rocprofv3 kernel stats of the seeds workload for the in-tree library and for $1 (a variant library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in product "$@"; do
  if [ "$L" = product ]; then unset SVOH_LIB; N=product; else export SVOH_LIB=$R/$L; N=$(basename $L .so); fi
  rm -rf /tmp/prof_$N
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -- python bench.py --workload seeds --steps 20 --warmup 3 --no-cpu-baseline > /tmp/prof_$N.log 2>&1 || tail -5 /tmp/prof_$N.log)
  F=$(find /tmp/prof_$N -name "*kernel_stats.csv" | head -1)
  echo "== $N"; python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'seed' in n or 'packed' in n:
        print('%-60s calls %5s avg_us %9.2f' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
