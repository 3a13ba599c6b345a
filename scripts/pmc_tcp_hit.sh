set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rocprofv3 --list-avail 2>/dev/null | grep -i -E "^\s*(Name|.*TCP_(TOTAL_CACHE|TCC_READ_REQ|TOTAL_ACCESS|PENDING|TA_).*)" | head -40 > gpurun_out/r06/tcp_counters.txt || true
rm -rf /tmp/pmc_l1 && rocprofv3 --pmc ${PMC_SET:-TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum} --kernel-include-regex "sparse_align" --output-format csv -d /tmp/pmc_l1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/pmc_l1.log 2>&1 || { tail -5 /tmp/pmc_l1.log; }
f=$(find /tmp/pmc_l1 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
