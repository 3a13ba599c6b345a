#!/bin/bash
# Runs on the GPU box.  For every benchmark workload: rocprofv3 kernel-trace/stats of the bench command, then separate
# counter passes of the SAME command (FETCH_SIZE, WRITE_SIZE, two SQ sets; never mixed with trace flags), raw CSVs of
# the svoh kernels kept, and one summary json per workload (scripts/pmc_summary.py) -> gpurun_out/profiles/.
# usage: scripts/profile_round.sh <round> [tags...]     tags: align_p4 align_p8 align_p4_b1024 align_p8_b1024 align_c4 klt seeds seeds_ws pose stereo
set -e
ROUND=${1:-r03}; shift || true
TAGS=${@:-align_p4 align_p8 align_c4 klt seeds pose stereo}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles
mkdir -p $dst
STEPS="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
for tag in $TAGS; do
  case $tag in
    align_p4) args="$STEPS"; key="align:B4096:N2000:P4:L4-0"; rx="sparse_align_kernel<4, 256, false, false|sparse_align_kernel.*Li4ELi256ELb0ELb0";;
    align_p8) args="--patch 8 $STEPS"; key="align:B4096:N2000:P8:L4-0"; rx="sparse_align_kernel<8, 256, false, false";;
    align_p4_b1024) args="--problems 1024 $STEPS"; key="align:B1024:N2000:P4:L4-0"; rx="sparse_align_kernel<4, 256, false, false";;   # the step of rounds 2 - 5 (two problems per resident workgroup)
    align_p8_b1024) args="--problems 1024 --patch 8 $STEPS"; key="align:B1024:N2000:P8:L4-0"; rx="sparse_align_kernel<8, 256, false, false";;
    align_c4) args="--workload align-c4 $STEPS"; key="align-c4:default"; rx="sparse_align_kernel<4, 256, true, false";;
    klt) args="--workload klt $STEPS"; key="klt:default"; rx="klt_track_kernel";;
    seeds) args="--workload seeds $STEPS"; key="seeds:default"; rx="update_seeds|seed_bin|seed_unsort";;
    seeds_ws) args="--workload seeds --whole-sets $STEPS"; key="seeds-ws:default"; rx="update_seeds_packed_kernel<true>";;   # (the one host-array call that checks the results brings its own bin kernels: not part of this step)
    pose) args="--workload pose $STEPS"; key="pose:default"; rx="pose_optimize_kernel";;
    stereo) args="--workload stereo $STEPS"; key="stereo:default"; rx="epipolar_match_kernel";;
  esac
  out=/tmp/prof_${ROUND}_$tag
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py $args > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
  grep '^{' $out/trace.log > $dst/${ROUND}_${tag}_bench_under_rocprof.json || true
  st=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  (head -1 $st; grep -E "svoh::" $st) > $dst/${ROUND}_${tag}_kernel_stats_svoh.csv
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32"; do
    t=$(echo $ctr | cut -d' ' -f1)
    [ "$t" = "SQ_WAVES" ] && t=SQ1; [ "$t" = "SQ_INSTS_VMEM_RD" ] && t=SQ2; [ "$t" = "SQ_INSTS_VALU_ADD_F64" ] && t=SQ3
    rocprofv3 --pmc $ctr --kernel-include-regex "svoh" --output-format csv -d $out/$t -- python bench.py $args > $out/$t.log 2>&1 || { tail -5 $out/$t.log; continue; }
    f=$(find $out/$t -name "*counter_collection.csv" | head -1)
    # keep the columns that matter, svoh kernels only (raw per-dispatch values: the summary can be re-derived)
    python - "$f" "$dst/${ROUND}_${tag}_pmc_$t.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value")
keep = [k for k in keep if rows and k in rows[0]]
w = csv.DictWriter(open(sys.argv[2], "w"), keep)
w.writeheader()
for r in rows:
    if "svoh" in r["Kernel_Name"]:
        r = {k: r[k] for k in keep}
        r["Kernel_Name"] = r["Kernel_Name"][:160]
        w.writerow(r)
PY
  done
  python scripts/pmc_summary.py $dst $ROUND $tag "$key" "$rx" $dst/${ROUND}_${tag}_bench_under_rocprof.json
done
echo profile_round done
