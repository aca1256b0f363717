#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/profile_r02; mkdir -p $OUT; export TMPDIR=/tmp
export DCTTS_SYNC_VALUES=0
cd /tmp
DM=3 GM=0 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_fetch.log" 2>&1
DM=3 GM=0 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_write.log" 2>&1
DM=3 GM=0 timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_sq.log" 2>&1
cd "$R"
python tools/pmc_summary.py "$OUT/pmc_fetch" FETCH_SIZE > "$OUT/pmc_fetch.txt"
python tools/pmc_summary.py "$OUT/pmc_write" WRITE_SIZE > "$OUT/pmc_write.txt"
for ctr in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; do
  echo "== $ctr"; python tools/pmc_summary.py "$OUT/pmc_sq" $ctr | head -12; done > "$OUT/pmc_sq.txt"
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
head -6 $OUT/pmc_fetch.txt; head -6 $OUT/pmc_write.txt
