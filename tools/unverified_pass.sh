#!/bin/bash
# Everything that was written after round 2's GPU minutes were spent, in ONE short GPU call (each step under its own timeout):
#   1. the opt-in GPU tests (training loop: 6 steps == 4 + resume + 2; waves -> prepo -> batches -> steps; the BPF = 2 / pinned / V2 kernels bitwise)
#   2. A/B of DCTTS_HCONV_BPF=1 vs 2: TextEnc and SSRN alone, then the bench line without extras
# Usage: gpurun --timeout 420 -- 'bash tools/unverified_pass.sh'
set -u
R=$PWD; OUT=$R/gpurun_out/unverified; mkdir -p $OUT
DCTTS_TEST_UNVERIFIED=1 timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -x -k "hconv_deeper or training_loop" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for v in 1 2 1 2; do
  echo "== DCTTS_HCONV_BPF=$v"
  DCTTS_HCONV_BPF=$v timeout 60 python tools/ssrn_time.py 32 2>&1 | grep "B="
  DCTTS_HCONV_BPF=$v timeout 90 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vocoder --no-extras 2>/dev/null | cut -c1-140
done | tee $OUT/ab.txt
