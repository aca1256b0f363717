#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged or long_form" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for mlp in 0 1; do for gm in 0 1; do
  DCTTS_MLP=$mlp DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_m${mlp}_gm${gm}.log 2>&1
  echo "MLP=$mlp GM=${gm}: $(grep text2mel $OUT/time_m${mlp}_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_m${mlp}_gm${gm}.log)"
done; done
DCTTS_V3_SKIP=1 DM=3 GM=0 timeout 120 python tools/decode_time.py 2>&1 | grep text2mel
cd /tmp
DM=3 GM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
find $OUT/kt3 -name "*kernel_stats.csv" | head -1 | xargs grep -i "mlp_rows\|attnq\|chain3_kernel<2, true, false"
