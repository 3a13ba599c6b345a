#!/bin/bash
# SQ counters of the 1024 x 2000 batch with 1 / 2 / 4 lanes per patch (512-thread geometry), and the kernel trace of the
# single 180-patch problem: the evidence behind DESIGN.md 4.2b.  Runs on the GPU box; writes gpurun_out/profiles/r04_*.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles; mkdir -p $dst
for R in 0 2 4; do
  out=/tmp/prof_rows_$R; rm -rf $out; mkdir -p $out
  export SVOH_ALIGN_ROWS=$R SVOH_ALIGN_THREADS=512 ILLUM=0
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-include-regex "sparse_align_kernel" --output-format csv -d $out/pmc -- python scripts/perf_quick.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" "$dst/r04_align_rows_lpp${R}_pmc_SQ1.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [k for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value") if rows and k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep); w.writeheader()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    r2 = {k: r[k] for k in keep}; r2["Kernel_Name"] = r2["Kernel_Name"][:120]; w.writerow(r2)
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    big = max(d["SQ_INSTS_VALU"])
    sel = [i for i, v in enumerate(d["SQ_INSTS_VALU"]) if v > 0.9 * big]   # the levels 4..0 launches of the script
    m = {c: sum(d[c][i] for i in sel) / len(sel) for c in d}
    print("%s: %d launches of levels 4..0: VALU wave-instructions %.3e, waves %d, VALU busy %.2f of SIMD time, waves waiting %.2f of their cycles" % (
        k[-60:], len(sel), m["SQ_INSTS_VALU"], m["SQ_WAVES"], m["SQ_ACTIVE_INST_VALU"] * 4 / (m["SQ_BUSY_CYCLES"] / 8 * 4 * 256 / 8) if False else m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] * (m["SQ_WAVE_CYCLES"] / (m["SQ_BUSY_CYCLES"])) / 4.0 if False else 0.0, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
    print("   raw means:", {c: "%.4g" % v for c, v in m.items()})
PY
done
unset SVOH_ALIGN_ROWS SVOH_ALIGN_THREADS
out=/tmp/prof_c3; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python scripts/perf_c3.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
st=$(find $out/trace -name "*kernel_stats.csv" | head -1)
(head -1 $st; grep -E "svoh::" $st) > $dst/r04_align_c3_single_problem_kernel_stats.csv
cat $dst/r04_align_c3_single_problem_kernel_stats.csv | cut -c1-200
grep kernel $out/log.txt
echo done
