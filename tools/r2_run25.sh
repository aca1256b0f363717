#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for dm in 3 4; do
  DM=$dm GM=1 timeout 120 python tools/decode_time.py > $OUT/time_dm$dm.log 2>&1
  echo "mode $dm: $(grep text2mel $OUT/time_dm$dm.log)"
done
