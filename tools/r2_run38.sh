#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_meeting or decode_vs_oracle_loop" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/sv.log 2>&1
echo "default: $(grep text2mel $OUT/sv.log)"
