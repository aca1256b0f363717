set -u
R=$PWD; OUT=$R/gpurun_out/r20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "decode_vs_oracle or end_of_text or large_batch or golden_config1" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2 3; do GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
GM=0 DCTTS_TRACE=150 DCTTS_TRACE_FILE=$OUT/decode_trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
cut -c1-330 $OUT/decode_trace.txt
