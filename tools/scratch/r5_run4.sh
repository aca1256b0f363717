set -u
R=$PWD; OUT=$R/gpurun_out/r4; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "DCTTS_XC_BD=2" "DCTTS_XC_BD=3" "DCTTS_XC_BD=4"; do
    echo "== $cfg" | tee -a $OUT/ssrn.txt
    env $cfg timeout 120 python tools/ssrn_time.py 32 2>&1 | grep SSRN | tee -a $OUT/ssrn.txt
  done
done
for cfg in "DCTTS_XC_BD=2" "DCTTS_XC_BD=3"; do
(cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/layers" -- python "$R/tools/layer_trace.py" > "$OUT/layers.log" 2>&1)
python tools/layer_trace_table.py "$OUT/layers" > "$OUT/layers_$cfg.txt"; rm -rf "$OUT/layers"
done
