set -u
R=$PWD; OUT=$R/gpurun_out/r8; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log; tail -5 $OUT/pytest_all.log
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r8/bench.json"))
print(d["value"], d["ms_per_step"], d["phases"], d["roofline"]["frac"])
for k,v in d["other_configs"].items(): print(k, json.dumps({a:b for a,b in v.items() if a not in("workload","arithmetic")})[:600])
PY
