set -u
R=$PWD; OUT=$R/gpurun_out/xg; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "not two_ranks and not bench_two and not long_form" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 200 python tools/ssrn_time.py 32 128 2>&1 | grep -E "SSRN|rror" | tee $OUT/ssrn.txt
