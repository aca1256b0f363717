set -u
R=$PWD; OUT=$R/gpurun_out/xg; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "layer or ssrn or golden or full_size or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 200 python tools/ssrn_time.py 32 128 2>&1 | grep -E "SSRN|rror" | tee $OUT/ssrn.txt
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/layers -- python $R/tools/layer_trace.py > $OUT/layers.log 2>&1); python tools/layer_trace_table.py $OUT/layers 2>/dev/null | tail -22 | tee $OUT/layers.txt; rm -rf $OUT/layers
