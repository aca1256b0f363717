import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
T = 6
h = hp.replace(max_T=T)
eng = Engine(synthetic_weights(h), h, decode_graph=0)
L = torch.from_numpy(synthetic_text(h, B=32)).cuda()
Y, mx = eng.text2mel(L)
torch.cuda.synchronize()
try:
    eng.decode_status(); print("status ok")
except Exception as e:
    print("status:", e)
os.environ["X"]="1"
e0 = Engine(synthetic_weights(h), h, decode_graph=0)
