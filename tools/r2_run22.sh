#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for rc in 0 1; do
  DCTTS_ROWCHAIN=$rc DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/time_rc$rc.log 2>&1
  echo "ROWCHAIN=$rc: $(grep text2mel $OUT/time_rc$rc.log)"
done
DCTTS_ROWCHAIN=1 DCTTS_V3_SKIP=1 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/skip1.log 2>&1
echo "chain only: $(grep text2mel $OUT/skip1.log)"
DCTTS_ROWCHAIN=1 DCTTS_V3_SKIP=2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/skip2.log 2>&1
echo "bulk only: $(grep text2mel $OUT/skip2.log)"
