#!/bin/bash
# Reproduces everything under profiles/ (round 6) on a 1x MI355X box (run from the repo root; ~5 minutes of GPU time).
# Every profiler command is wrapped in `timeout`; --pmc passes are separate runs with --kernel-trace only.
set -u
R=$PWD
OUT=${1:-$R/gpurun_out/profile_r06}
mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS=${STEPS:-012345789}           # e.g. STEPS=127 bash tools/profile_all.sh: only the bench line, the kernel trace and the layer table
want() { case "$STEPS" in *$1*) return 0;; *) return 1;; esac; }
# 1. the bench line (metric, roofline of the time-dominant kernel, kernels, other configs, cpu_baseline, vocoder)  -> profiles/r06_bench.json
want 1 && timeout 600 python "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp
# 2. per-kernel time of the same workload                                               -> profiles/r06_kernel_stats.{csv,md}
want 2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernel_stats" -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-vocoder --no-extras > "$OUT/kernel_stats.log" 2>&1
# 3. HBM traffic of the decode kernels (FETCH_SIZE x2 per the gfx950 correction; WRITE_SIZE exact)   -> profiles/r06_pmc.{md,json}
#    Counter collection serialises dispatches ACROSS queues, so a launch that waits for the other stream's counter (stream memory
#    operations, in-kernel signals) would wait for ever: the counter passes let the two streams meet through events (DCTTS_SYNC_VALUES=0).
export DCTTS_SYNC_VALUES=0
want 3 && DM=3 GM=0 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_fetch.log" 2>&1
want 3 && DM=3 GM=0 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_write.log" 2>&1
# 4. what the waves of the decode kernels wait for
want 4 && DM=3 GM=0 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -- python "$R/tools/decode_only.py" 40 > "$OUT/pmc_sq.log" 2>&1
unset DCTTS_SYNC_VALUES
cd "$R"
want 3 && python tools/pmc_summary.py "$OUT/pmc_fetch" FETCH_SIZE > "$OUT/pmc_fetch.txt"
want 3 && python tools/pmc_summary.py "$OUT/pmc_write" WRITE_SIZE > "$OUT/pmc_write.txt"
want 3 && python tools/pmc_json.py "$OUT/pmc_fetch.txt" "$OUT/pmc_write.txt" "$OUT/pmc_decode.json" > /dev/null
want 4 && for ctr in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; do
  echo "== $ctr"; python tools/pmc_summary.py "$OUT/pmc_sq" $ctr | head -12; done > "$OUT/pmc_sq.txt"
# 5. where the team kernels / mlp_rows_kernel spend their time (in-kernel wall-clock stamps) and how long the two streams' pieces take
want 5 && DCTTS_TRACE_FILE="$OUT/decode_trace.txt" timeout 100 python "$R/tools/decode_trace.py" > "$OUT/decode_trace.log" 2>&1
want 5 && DCTTS_PIECETIME=100 DM=3 GM=0 timeout 120 python "$R/tools/decode_time.py" > "$OUT/piece_times.txt" 2>&1
# 7. TextEnc and SSRN launch by launch
want 7 && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/layers" -- python "$R/tools/layer_trace.py" > "$OUT/layers.log" 2>&1)
want 7 && python tools/layer_trace_table.py "$OUT/layers" > "$OUT/layers.txt"
# 8. the throughput kernel's variants on four layer shapes, each timed in turn behind a cache-thrashing pass (product template only)   -> profiles/r06_hconv_lab.txt
want 8 && (hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dc_tts_amd/csrc tools/micro/hconv_lab.hip -o tools/micro/kp_hconv_lab 2> "$OUT/hconv_lab_build.log"; timeout 300 tools/micro/kp_hconv_lab 9 > "$OUT/hconv_lab.txt" 2>&1)
# 9. what the two decode streams' pieces take in the default form (2: four newest-row layers on the chain, HC_5's older rows on the side stream), with round 5's split of the cone (7), as two launches per chain piece (6), and with HC_2..HC_4 as a launch of their own (5)      -> profiles/r06_chain_tail_split.txt
want 9 && for tail in 2 7 6 5; do
  echo "== DCTTS_CHAIN_TAIL=$tail"
  for rep in 1 2; do DCTTS_CHAIN_TAIL=$tail GM=0 HP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "text2mel|rror"; done
  DCTTS_CHAIN_TAIL=$tail DCTTS_PIECETIME=150 GM=0 HP=1 NREP=1 timeout 120 python tools/decode_time.py 2>&1 | grep -E "frame 15[0-7]" | tail -8
done > "$OUT/chain_tail_split.txt" 2>&1
# ... and the two parts of the one-launch chain piece timed as launches of their own (DCTTS_CHAIN_TAIL=6: xtail_kernel + xgroup_kernel, HIP events, bench.py's kernels[])
want 9 && (echo "== DCTTS_CHAIN_TAIL=6: bench.py kernels[] (xtail_kernel, xgroup_kernel: avg_launch_ms)"; DCTTS_CHAIN_TAIL=6 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vocoder 2>/dev/null | \
  python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); [print(k['kernel'][:60], '| launches', k['launches'], '| avg_launch_ms', k['avg_launch_ms']) for k in j.get('kernels', []) if 'decode chain' in k['kernel']]; print('decode_us_per_step', j['phases']['decode_us_per_step'])") >> "$OUT/chain_tail_split.txt" 2>&1
# 10. the MFMA feed lab (what a wave keeps of the matrix pipe when its operands arrive during the loop) and the first-decode / stream-pair probes
want 0 && (hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_feed_lab.hip -o tools/micro/kp_mfma_feed_lab 2> "$OUT/mfma_feed_lab_build.log"; timeout 120 tools/micro/kp_mfma_feed_lab > "$OUT/mfma_feed_lab.txt" 2>&1)
want 0 && (for ct in 2 7 6 5 1 0; do CT=$ct REPS=1 BS=1,3,5,8,12,32 TS=60,70,80,90 timeout 500 python tools/flaky_probe.py 2>&1 | grep -v amdgpu.ids | tail -3; done; for k in "HI=1" "HI=1 DCTTS_CHAIN_WAIT=0" "HI=1 DCTTS_XGROUP=0 DCTTS_XCONE=0"; do echo "ctx_probe $k: $(env $k NNEW=6 timeout 200 python tools/ctx_probe.py 2>&1 | tail -1)"; done) > "$OUT/first_decode_probe.txt" 2>&1
want 0 && (for xg in 1 2; do XG=$xg timeout 200 python tools/config5_time.py 2>&1 | tail -1; done; for b in 16 4; do for xg in 1 2; do XG=$xg B5=$b T5=210 timeout 200 python tools/config5_time.py 2>&1 | tail -1; done; done) > "$OUT/config5_teams.txt" 2>&1
want 2 && find "$OUT/kernel_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
rm -rf "$OUT/kernel_stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/layers"
echo "done: $OUT"
