"""One TextEnc call and one SSRN call (B = 32, LJ shapes) under `rocprofv3 --kernel-trace`: tools/layer_trace_table.py turns the trace
into a per-launch table (kernel, duration, TFLOP/s is derived there from the layer list)."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L
if os.environ.get("DCTTS_AB_LIB"): _L.LIB_PATH = os.environ["DCTTS_AB_LIB"]      # A/B of two builds of the library (tools only)
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
Y = torch.rand(32, hp.max_T, hp.n_mels, device="cuda")
for _ in range(2):
    eng.text_enc(L); eng.ssrn(Y)
torch.cuda.synchronize()
