# TextEnc alone: time per call (B = 32, N = 180)
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L0
if os.environ.get("DCTTS_AB_LIB"): _L0.LIB_PATH = os.environ["DCTTS_AB_LIB"]
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
for _ in range(3): eng.text_enc(L)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20): eng.text_enc(L)
t1.record(); torch.cuda.synchronize()
print("TextEnc ms", t0.elapsed_time(t1) / 20)
