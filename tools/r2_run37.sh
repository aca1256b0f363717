#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for g in 0 1; do for dm in 3 4; do
  DCTTS_GATE=$g DM=$dm GM=1 timeout 120 python tools/decode_time.py > $OUT/sv.log 2>&1
  echo "gate $g mode $dm: $(grep text2mel $OUT/sv.log)"
done; done
DCTTS_GATE=1 DM=3 GM=2 timeout 120 python tools/decode_time.py > $OUT/sv.log 2>&1
echo "gate 1 mode 3 chain graphs: $(grep text2mel $OUT/sv.log)"
