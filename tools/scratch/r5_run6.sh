set -u
R=$PWD; OUT=$R/gpurun_out/r6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "split_bf16" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -E "max-abs|passed|failed|Error|error|assert" $OUT/pytest.log | head -60
python - <<'PY' 2>&1 | tee $OUT/bf_time.txt
import torch, sys
sys.path.insert(0, '.')
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp, seed=1), hp, split_bf16=2)
Y = torch.rand(32, 210, hp.n_mels, device="cuda"); L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for mode in (0, 1, 2):
    eng.set_split_bf16(mode)
    print(f"mode {mode}: SSRN {t(lambda: eng.ssrn(Y, want_logits=False)):.3f} ms, TextEnc {t(lambda: eng.text_enc(L)):.3f} ms, synthesize {t(lambda: eng.synthesize(L), 5):.3f} ms")
PY
