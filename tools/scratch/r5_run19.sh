set -u
R=$PWD; OUT=$R/gpurun_out/r19; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python tools/soak.py --n1 1500 --n2 200 --n3 500 --n4 200 > $OUT/soak.txt 2>&1; tail -6 $OUT/soak.txt
