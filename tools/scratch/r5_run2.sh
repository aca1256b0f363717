set -u
R=$PWD; OUT=$R/gpurun_out/r2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "one_engine or alternating or checked or two_contexts or high_priority or failed_decode or survives or decode_vs_oracle or ragged or large_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 300 python tools/soak.py --n1 200 --n2 40 --n3 100 --n4 100 --every 50 > $OUT/soak_short.txt 2>&1; tail -8 $OUT/soak_short.txt
for rep in 1 2; do GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done | tee $OUT/dt.txt
