set -u
R=$PWD; OUT=$R/gpurun_out/r7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "split_bf16" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -E "max-abs|passed|failed|Error|error|assert" $OUT/pytest.log | head -60
cat > /tmp/bf_trace.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp, split_bf16=2)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
Y = torch.rand(32, hp.max_T, hp.n_mels, device="cuda")
for _ in range(2):
    eng.text_enc(L); eng.ssrn(Y)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/layers" -- python /tmp/bf_trace.py > "$OUT/layers.log" 2>&1)
python tools/layer_trace_table.py "$OUT/layers" > "$OUT/layers_bf16.txt"; rm -rf "$OUT/layers"
tools/micro/kp_hconv_lab 7 > $OUT/lab.txt 2>&1; tail -40 $OUT/lab.txt
