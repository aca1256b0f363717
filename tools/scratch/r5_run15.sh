set -u
R=$PWD; OUT=$R/gpurun_out/r15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "decode_vs_oracle or end_of_text or large_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do
for cfg in "DCTTS_XCONE=1" "DCTTS_XCONE=2"; do echo "== $cfg"; env $cfg GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
done
GM=0 DCTTS_TRACE=150 DCTTS_TRACE_FILE=$OUT/decode_trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
grep -A1 "xcone_kernel" $OUT/decode_trace.txt | cut -c1-300
