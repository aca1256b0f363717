#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for sk in 0 1 2; do for gm in 0 1 2; do
  DCTTS_V3_SKIP=${sk} DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_skip${sk}_gm${gm}.log 2>&1
  echo "SKIP=${sk} GM=${gm}: $(grep text2mel $OUT/time_skip${sk}_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_skip${sk}_gm${gm}.log)"
done; done
cd /tmp
DM=3 GM=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
find $OUT/kt3 -name "*kernel_stats.csv" | head -1 | xargs head -14
