#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>"   -> gpurun_out/pmc_<tag>.csv (only the alignment kernel rows)
set -e
tag=$1; ctrs=$2
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-include-regex "${KREGEX:-sparse_align}" --output-format csv -d $out -- python ${PROG:-scripts/prof_align.py} > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "$tag" <<'PY'
import sys, csv, collections
f, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("PMC", tag, {k: sum(v) / len(v) for k, v in acc.items()}, "dispatches", len(rows) // max(1, len(acc)))
PY
