import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
from oracle import dctts_ref as O
T = int(os.environ.get("TT", "30"))
h = hp.replace(max_T=T)
W = synthetic_weights(hp, seed=1234, perturb=True)
eng = Engine(W, h)
prev0 = np.array([170, 174, 176, 177, 178, 179], np.int32)[: int(os.environ.get("BB", "6"))]
L = synthetic_text(h, B=len(prev0), seed=5)
Yr, _, trajr = O.synthesize(L, W, h, np.float32, run_ssrn=False, prev0=prev0)
for rep in range(3):
    eng.debug_seed_prev_max(prev0)
    Y, mx = eng.text2mel(torch.from_numpy(L).cuda())
    eng.synchronize()
    e = np.abs(Y.cpu().numpy() - Yr)            # (B, T, 80)
    print("rep", rep, "traj ok", (mx.cpu().numpy() == trajr).all(), "max err", e.max())
    print("  per utterance:", np.round(e.max(axis=(1, 2)), 4))
    print("  per frame    :", np.round(e.max(axis=(0, 2)), 3))
