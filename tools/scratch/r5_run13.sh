set -u
R=$PWD; OUT=$R/gpurun_out/r13; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "decode_vs_oracle or stream_meeting or end_of_text or golden_config1 or large_batch or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for rep in 1 2; do
for cfg in "DCTTS_XCONE=1" "DCTTS_XCONE=2"; do echo "== $cfg"; env $cfg GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
done
DCTTS_PIECETIME=150 GM=0 HP=1 NREP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "frame 15[0-7]" | tail -3
