import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
W = synthetic_weights(hp)
eng = Engine(W, hp, decode_graph=int(os.environ.get("GM", "0")))
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
eng.text2mel(L); torch.cuda.synchronize()
out = os.environ.setdefault('DCTTS_TRACE_FILE', 'gpurun_out/decode_trace.txt')
os.environ['DCTTS_TRACE'] = '150'
eng.text2mel(L); torch.cuda.synchronize()
print(open(out).read())
