import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L0
if os.environ.get("DCTTS_AB_LIB"): _L0.LIB_PATH = os.environ["DCTTS_AB_LIB"]
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
T = 210
eng = Engine(synthetic_weights(hp), hp, decode_graph=int(os.environ.get("GM", "1")))
eng.set_decode_mode(int(os.environ.get("DM", "3")))
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
if os.environ.get("HP"): torch.cuda.set_stream(torch.cuda.Stream(priority=-1))      # a high-priority caller's stream: the decode's chain runs on it directly
for _ in range(2): eng.text2mel(L)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5): eng.text_enc(L)
t1.record(); torch.cuda.synchronize()
TE = t0.elapsed_time(t1) / 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
NREP = int(os.environ.get("NREP", "3"))
for _ in range(NREP): eng.text2mel(L)
e1.record(); torch.cuda.synchronize()
print("text2mel ms", e0.elapsed_time(e1) / NREP, "us/frame", (e0.elapsed_time(e1) / NREP - TE) * 1e3 / T, "(TextEnc %.3f ms subtracted)" % TE)
