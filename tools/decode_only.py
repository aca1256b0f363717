import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L0
if os.environ.get("DCTTS_AB_LIB"): _L0.LIB_PATH = os.environ["DCTTS_AB_LIB"]
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
h = hp.replace(max_T=T)
eng = Engine(synthetic_weights(h), h, decode_graph=int(os.environ.get("GM", "0")))
eng.set_decode_mode(int(os.environ.get("DM", "3")))
L = torch.from_numpy(synthetic_text(h, B=32)).cuda()
eng.text2mel(L); torch.cuda.synchronize()
eng.text2mel(L); torch.cuda.synchronize()
print("done")
