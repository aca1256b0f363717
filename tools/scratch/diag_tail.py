import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
from oracle import dctts_ref as O
T = int(os.environ.get("T", "70")); B = int(os.environ.get("B", "3"))
h = hp.replace(max_T=T)
W = synthetic_weights(h, seed=1234, perturb=True)
Lh = synthetic_text(h, B=B, seed=21)
Yr, _, trajr = O.synthesize(Lh, W, h, np.float32, run_ssrn=False)
res = {}
for v in ("2", "1"):
    os.environ["DCTTS_CHAIN_TAIL"] = v
    eng = Engine(W, h)
    Y, mx = eng.text2mel(torch.from_numpy(Lh).cuda())
    eng.synchronize()
    Y = Y.cpu().numpy(); mx = mx.cpu().numpy()
    err = np.abs(Y - Yr).max(axis=(0, 2))
    bad = np.flatnonzero((mx != trajr).any(0))
    print("CHAIN_TAIL", v, "first traj mismatch frame", bad[:1], "err per frame:", " ".join("%.1e" % e for e in err[:60]))
    res[v] = Y
    eng.close()
d = np.abs(res["2"] - res["1"]).max(axis=(0, 2))
print("tail2 vs tail1 per frame:", " ".join("%.1e" % e for e in d[:60]))
