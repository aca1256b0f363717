#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for cfg in "16 176" "16 144" "16 208" "16 256" "6 176" "46 176"; do set -- $cfg
  DCTTS_HOSTTIME=1 DCTTS_BULK3_SMALL=$1 DCTTS_BULK_CAP=$2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/time_s$1_c$2.log 2>&1
  echo "SMALL=$1 CAP=$2: $(grep text2mel $OUT/time_s$1_c$2.log) $(grep -m1 'host enqueue' $OUT/time_s$1_c$2.log)"
done
cd /tmp
DM=3 GM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
