#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for cfg in "0 0" "0 512" "1 512" "1 0"; do set -- $cfg
  DCTTS_BULK3_FUSED=$1 DCTTS_BULK3_SMALL=$2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/time_f$1_s$2.log 2>&1
  echo "FUSED=$1 SMALL=$2: $(grep text2mel $OUT/time_f$1_s$2.log)"
done
cd /tmp
DM=3 GM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
find $OUT/kt3 -name "*kernel_stats.csv" | head -1 | xargs head -12
