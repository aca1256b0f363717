#!/bin/bash
# Round-end GPU pass: the whole GPU test suite, smoke(), and everything under profiles/ (tools/profile_all.sh).
set -u
R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
STEPS=${STEPS:-012345789} bash tools/profile_all.sh $R/gpurun_out/profile_r06 > $OUT/profile_all.log 2>&1; tail -2 $OUT/profile_all.log
bash tools/decode_probe.sh > $R/gpurun_out/profile_r06/decode_probe.txt 2>&1
timeout 900 python tools/soak.py --n1 5000 --n2 500 --n3 1500 --n4 1000 > $R/gpurun_out/profile_r06/soak.txt 2>&1; tail -5 $R/gpurun_out/profile_r06/soak.txt
head -c 600 $R/gpurun_out/profile_r06/bench.json; echo
