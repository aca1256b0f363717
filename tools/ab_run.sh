set -u
R=$PWD; OUT=$R/gpurun_out/xg; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "decode or end_of_text or golden or long_form" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2 3; do GM=0 timeout 100 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done | tee $OUT/ab6.txt
GM=0 DCTTS_TRACE=150 DCTTS_TRACE_FILE=gpurun_out/decode_trace.txt timeout 100 python tools/decode_trace.py 2>&1 | grep -A1 "xcone_kernel"
