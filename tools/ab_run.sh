# A/B of two builds of the library (dc_tts_amd/lib/libdctts_hip_base.so = the build before a change, libdctts_hip.so = the current one):
# parity tests of the current build, SSRN / TextEnc phase times of both, per-launch tables of both.
set -u
R=$PWD; OUT=$R/gpurun_out/ab; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x  > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export DCTTS_AB_LIB=$R/dc_tts_amd/lib/libdctts_hip_base.so; else unset DCTTS_AB_LIB; fi
  echo "== $lib"; timeout 100 python tools/ssrn_time.py 32 128 2>&1 | grep "B="
done; done | tee $OUT/ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in base new; do
  if [ $lib = base ]; then export DCTTS_AB_LIB=$R/dc_tts_amd/lib/libdctts_hip_base.so; else unset DCTTS_AB_LIB; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/layers_$lib -- python $R/tools/layer_trace.py > $OUT/layers_$lib.log 2>&1
  python $R/tools/layer_trace_table.py $OUT/layers_$lib > $OUT/layers_$lib.txt
done
paste $OUT/layers_base.txt $OUT/layers_new.txt | cut -c1-60,100-160 | awk 'NR>1'
