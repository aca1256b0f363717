#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
for grp in 0 1; do for gm in 0 1 2; do
  DCTTS_GROUP=$grp DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_g${grp}_gm${gm}.log 2>&1
  echo "GROUP=$grp GM=${gm}: $(grep text2mel $OUT/time_g${grp}_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_g${grp}_gm${gm}.log)"
done; done
DCTTS_V3_SKIP=1 DM=3 GM=0 timeout 120 python tools/decode_time.py 2>&1 | grep text2mel
