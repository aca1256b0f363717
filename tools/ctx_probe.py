"""A decode from a high-priority caller's stream + SSRN on a second stream on engine 1 (tests/test_gpu_parity.py::test_decode_from_a_high_priority_stream_...), then the FIRST
decode of a new engine from the default stream: the pair that failed in round 6 (error word 36).  DCTTS_AB_LIB selects another build."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L0
if os.environ.get('DCTTS_AB_LIB'): _L0.LIB_PATH = os.environ['DCTTS_AB_LIB']
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
W = synthetic_weights(hp, seed=1234, perturb=True)
T = 60
h = hp.replace(max_T=T)
eng = Engine(W, h)
L = torch.from_numpy(synthetic_text(h, B=32, seed=31)).cuda()
Y, mx = eng.text2mel(L); Z = eng.ssrn(Y, want_logits=False)[1]; eng.synchronize()
if os.environ.get("HI", "1") == "1":
    hi = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(hi):
        Yh, mh = eng.text2mel(L)
    torch.cuda.synchronize()
if os.environ.get("S2", "1") == "1":
    s2 = torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(s2):
            Zs = [eng.ssrn(Y, want_logits=False)[1] for _ in range(3)]
        Y2, m2 = eng.text2mel(L)
        torch.cuda.synchronize(); eng.decode_status()
bad = 0
for k in range(int(os.environ.get("NNEW", "3"))):
    h2 = hp.replace(max_T=50 + k)
    e2 = Engine(W, h2)
    L2 = torch.from_numpy(synthetic_text(h2, B=4, seed=77)).cuda()
    try:
        e2.synthesize(L2); e2.synchronize()
    except Exception as ex:
        bad += 1; print("new engine", k, "FAILED:", str(ex)[:120])
    e2.close()
print("failed first decodes of new engines:", bad)
