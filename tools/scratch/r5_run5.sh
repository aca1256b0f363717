set -u
R=$PWD; OUT=$R/gpurun_out/r5; mkdir -p $OUT
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/bench.json"))
print(d["value"], d["ms_per_step"], d["phases"], d["roofline"]["frac"])
for k,v in d["other_configs"].items(): print(k, {a:b for a,b in v.items() if a!="workload"})
PY
