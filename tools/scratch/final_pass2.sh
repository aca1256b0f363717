#!/bin/bash
# Round-end GPU pass, second run (after the last decode changes): GPU suite, smoke, bench line, kernel stats, decode trace / piece times, the chain forms, decode probe, soak.
set -u
R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
STEPS=1259 bash tools/profile_all.sh $R/gpurun_out/profile_r05 > $OUT/profile_all.log 2>&1; tail -2 $OUT/profile_all.log
bash tools/decode_probe.sh > $R/gpurun_out/profile_r05/decode_probe.txt 2>&1
timeout 900 python tools/soak.py --n1 5000 --n2 500 --n3 1500 --n4 1000 > $R/gpurun_out/profile_r05/soak.txt 2>&1; tail -5 $R/gpurun_out/profile_r05/soak.txt
head -c 400 $R/gpurun_out/profile_r05/bench.json; echo
