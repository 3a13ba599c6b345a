#!/bin/bash
# usage: scripts/pmc_seeds.sh <tag> [G8 value]  -> gpurun_out/pmc_seeds_<tag>.txt  (SQ + cache counters of the seed-update kernel)
tag=$1; g=${2:-2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SVOH_MATCHER_G8=$g
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmcs_$t
  rocprofv3 --pmc $set --kernel-include-regex "update_seeds" --output-format csv -d /tmp/pmcs_$t -- python bench.py --workload seeds --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmcs_$t.log 2>&1 || { tail -5 /tmp/pmcs_$t.log; continue; }
  cat $(find /tmp/pmcs_$t -name "*counter_collection.csv" | head -1) >> /tmp/pmc_seeds_all.csv
done
python - $tag <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open("/tmp/pmc_seeds_all.csv")) if r.get("Counter_Name") not in (None, "Counter_Name")]
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_seeds_%s.txt" % sys.argv[1], "w") as f:
    for k, v in sorted(acc.items()):
        line = "%-42s %-30s n=%d mean=%.5g" % (k[0], k[1], len(v), sum(v) / len(v))
        print(line); f.write(line + "\n")
PY
rm -f /tmp/pmc_seeds_all.csv
