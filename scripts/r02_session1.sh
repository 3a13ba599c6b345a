#!/bin/bash
# Round-2 first GPU session: tests, calibration microbench (+FETCH_SIZE pass), counter list, compute-side counters
# of the alignment kernel at the default bench size.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02s1
mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; }
tail -3 $out/pytest_gpu.log
./tools/svoh_microbench 2048 > $out/microbench.json
cat $out/microbench.json
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
rm -rf /tmp/mb_fetch
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/mb_fetch -- ./tools/svoh_microbench 2048 > $out/microbench_under_pmc.json 2> $out/mb_fetch.log || { tail -20 $out/mb_fetch.log; exit 1; }
cp $(find /tmp/mb_fetch -name "*counter_collection.csv" | head -1) $out/microbench_pmc_FETCH_SIZE.csv
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $set --kernel-include-regex "sparse_align" --output-format csv -d /tmp/pmc_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/pmc_$tag.log 2>&1 || { tail -5 $out/pmc_$tag.log; continue; }
  cp $(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1) $out/align_p4_pmc_$tag.csv || true
done
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmcs_$tag
  rocprofv3 --pmc $set --kernel-include-regex "update_seeds" --output-format csv -d /tmp/pmcs_$tag -- python bench.py --workload seeds --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_seeds_$tag.log 2>&1 || { tail -5 $out/pmc_seeds_$tag.log; continue; }
  cp $(find /tmp/pmcs_$tag -name "*counter_collection.csv" | head -1) $out/seeds_pmc_$tag.csv || true
done
echo done
