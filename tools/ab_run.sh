# A/B of two builds of the library in ONE gpurun call (boxes differ by up to 3 us per frame): the decode parity tests on the tree's build, then
# decode_time.py on the tree's build and on $BASE (default dc_tts_amd/lib/libdctts_hip_base.so, a build of an earlier commit) in alternation, then the stamps.
# usage: [ALT_ENV="DCTTS_TAIL_NP=3"] bash tools/ab_run.sh [pytest -k expression]   (ALT_ENV: a third leg, the tree's build under other knobs)
set -u
R=$PWD; OUT=$R/gpurun_out/ab; mkdir -p $OUT
BASE=${BASE:-$R/dc_tts_amd/lib/libdctts_hip_base.so}
K=${1:-"decode or end_of_text or golden or long_form or large_batch or shard or team or stream_meeting"}
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests/test_gpu_parity.py -q -x -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for rep in 1 2 3; do
  echo "new:  $(GM=0 HP=1 timeout 100 python tools/decode_time.py 2>&1 | grep -E 'text2mel|rror')"
  [ -f $BASE ] && echo "base: $(DCTTS_AB_LIB=$BASE GM=0 HP=1 timeout 100 python tools/decode_time.py 2>&1 | grep -E 'text2mel|rror')"
  [ -n "${ALT_ENV:-}" ] && echo "alt ($ALT_ENV): $(env $ALT_ENV GM=0 HP=1 timeout 100 python tools/decode_time.py 2>&1 | grep -E 'text2mel|rror')"
done | tee $OUT/ab.txt
GM=0 HP=1 DCTTS_TRACE=150 DCTTS_TRACE_FILE=gpurun_out/ab/decode_trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1; cat $OUT/decode_trace.txt
GM=0 HP=1 DCTTS_PIECETIME=150 timeout 100 python tools/decode_time.py 2>&1 | grep -E "frame 15[1-4]|text2mel" | tee $OUT/piecetime.txt
