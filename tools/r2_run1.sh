#!/bin/bash
# round-2 GPU check 1: v3 decode parity + timing vs v2 + kernel trace
set -u
R=$PWD; OUT=$R/gpurun_out/r2a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "decode_vs_oracle_loop or end_of_text or validated or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for dm in 1 3; do for gm in 1; do
  DM=$dm GM=$gm DCTTS_HOSTTIME=1 timeout 120 python tools/decode_time.py > $OUT/time_dm${dm}_gm${gm}.log 2>&1
  echo "DM=$dm GM=$gm: $(grep text2mel $OUT/time_dm${dm}_gm${gm}.log) | $(grep -m1 'host enqueue' $OUT/time_dm${dm}_gm${gm}.log)"
done; done
DM=3 GM=0 timeout 120 python tools/decode_time.py > $OUT/time_dm3_gm0.log 2>&1; echo "DM=3 GM=0: $(grep text2mel $OUT/time_dm3_gm0.log)"
cd /tmp
DM=3 GM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
find $OUT/kt3 -name "*kernel_stats.csv" | head -1 | xargs head -25
