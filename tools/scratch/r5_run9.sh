set -u
R=$PWD; OUT=$R/gpurun_out/r9; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vocoder.py -q -x -k "new_geometry or alternating or one_engine or vocoder or large_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r9/bench.json"))
print(d["value"], d["ms_per_step"], d["phases"], d["roofline"]["frac"])
for k,v in d["other_configs"].items(): print(k, json.dumps({a:b for a,b in v.items() if a not in("workload","arithmetic")})[:700])
PY
