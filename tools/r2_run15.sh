#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -6 $OUT/pytest_all.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2o/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","rtf")}); print(d["phases"]); print(d["roofline"]); print(d["kernels"]); print(d["other_configs"]); print(d["cpu_baseline"]); print(d["gather"])
PY
tail -3 $OUT/bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 2 --warmup 1 --share-gpu --dist-backend gloo --no-extras > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?"; tail -c 600 $OUT/bench2.json; tail -3 $OUT/bench2.err
