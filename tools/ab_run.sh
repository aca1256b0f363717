set -u
R=$PWD; OUT=$R/gpurun_out/xg; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "decode_vs_oracle or end_of_text or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do echo "== xgroup+xcone GM=0"; GM=0 timeout 100 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done | tee $OUT/ab3.txt
GM=0 DCTTS_PIECETIME=100 timeout 100 python tools/decode_time.py 2>&1 | grep -E "frame 10[2-4]" | tee -a $OUT/ab3.txt
GM=0 bash tools/decode_probe.sh 2>&1 | head -12 | tee -a $OUT/ab3.txt
