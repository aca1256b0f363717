#!/bin/bash
# Reproduces everything under profiles/ on a 1x MI355X box (run from the repo root; ~4 minutes of GPU time).
# Every profiler command is wrapped in `timeout`: a rocprofv3 --pmc run with an unsupported counter set once aborted and hung.
# --pmc passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with the sys/hip/hsa trace domains).
set -u
R=$PWD
OUT=${1:-$R/gpurun_out/profile_all}
mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. the bench line (metric, roofline, cpu_baseline, vocoder)                       -> profiles/r01_bench_final.json
timeout 300 python "$R/bench.py" > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
cd /tmp
# 2. per-kernel time of the same command                                             -> profiles/r01_kernel_stats.{csv,md}
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_stats" -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder > "$OUT/kernel_stats.log" 2>&1
# 3. HBM traffic of the roofline kernel (FETCH_SIZE x2 per the gfx950 correction, calibrated by a 1 GiB copy; WRITE_SIZE exact)
#                                                                                    -> profiles/r01_pmc_traffic.{json,md}
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/tools/pmc_ssrn.py" > "$OUT/pmc_fetch.log" 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/tools/pmc_ssrn.py" > "$OUT/pmc_write.log" 2>&1
# 4. MFMA-busy of the SSRN kernels                                                   -> profiles/r01_pmc_traffic.md (second table)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -- python "$R/tools/pmc_ssrn.py" > "$OUT/pmc_sq.log" 2>&1
# 5. vocoder tail: kernel trace + HBM traffic                                        -> profiles/r01_vocoder*.{csv,md,json}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/voc_kernel_stats" -- python "$R/tools/vocoder_bench.py" > "$OUT/voc_kernel_stats.log" 2>&1
VREPS=1 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/voc_pmc_fetch" -- python "$R/tools/vocoder_bench.py" > "$OUT/voc_pmc_fetch.log" 2>&1
VREPS=1 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/voc_pmc_write" -- python "$R/tools/vocoder_bench.py" > "$OUT/voc_pmc_write.log" 2>&1
cd "$R"
# 6. where a decode chain launch spends its time (in-kernel wall-clock stamps)       -> gpurun_out/decode_trace.txt
timeout 100 python "$R/tools/decode_trace.py" > "$OUT/decode_trace.log" 2>&1
echo "done: $OUT"
