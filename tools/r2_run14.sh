#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_all.log
tail -8 $OUT/pytest_all.log
timeout 400 python bench.py --no-cpu-baseline --no-vocoder > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
