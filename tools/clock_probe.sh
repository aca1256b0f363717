# What clock / power does the chip run at under SSRN (fp32 MFMA at ~0.7 of peak) and under the decode?  rocm-smi sampled beside a looping workload.
set -u
R=$PWD; OUT=$R/gpurun_out/clock; mkdir -p $OUT
rocm-smi --showclocks --showpower --showtemp > $OUT/idle.txt 2>&1
rocm-smi --showmaxpower --showperflevel >> $OUT/idle.txt 2>&1
export OUTD=$OUT
( python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights
eng = Engine(synthetic_weights(hp, seed=1), hp)
Y = torch.rand(32, 210, hp.n_mels, device="cuda")
t0 = time.time()
open(os.environ["OUTD"] + "/started", "w").write("x")
n = 0
while time.time() - t0 < 8.0:
    for _ in range(20): eng.ssrn(Y, want_logits=False)
    torch.cuda.synchronize(); n += 20
print("ssrn calls", n, "avg ms", (time.time() - t0) / n * 1e3)
PY
) > $OUT/work.log 2>&1 &
WP=$!
export OUTD=$OUT
while [ ! -f $OUT/started ]; do sleep 0.2; done
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.4
done > $OUT/busy.txt
wait $WP
cat $OUT/work.log | tail -2; head -30 $OUT/idle.txt; cat $OUT/busy.txt
