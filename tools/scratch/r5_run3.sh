set -u
R=$PWD; OUT=$R/gpurun_out/r3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "ssrn or layer_row_split or (layer_vs_oracle and ssrn) or networks_surface" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for rep in 1 2; do
  for cfg in "DCTTS_SSRN_XC=0" "DCTTS_SSRN_XC=1 DCTTS_XC_BD=1" "DCTTS_SSRN_XC=1 DCTTS_XC_BD=2"; do
    echo "== $cfg" | tee -a $OUT/ssrn.txt
    env $cfg timeout 120 python tools/ssrn_time.py 32 128 2>&1 | grep SSRN | tee -a $OUT/ssrn.txt
  done
done
