"""PMC workload: a calibration copy of known size, then SSRN at the bench shape (B=32, T=210), twice."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights
eng = Engine(synthetic_weights(hp), hp)
n = 256 * 1024 * 1024            # 1 GiB of floats read + 1 GiB written
a = torch.rand(n, device='cuda'); b = torch.empty_like(a)
for _ in range(2):
    eng.lib.dctts_debug_copy(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, None)
torch.cuda.synchronize()
Y = torch.rand(32, 210, hp.n_mels, device='cuda')
for _ in range(2):
    eng.ssrn(Y, want_logits=False)
torch.cuda.synchronize()
print("done")
