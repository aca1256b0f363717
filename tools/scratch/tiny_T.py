"""Decode at max_T = 1, 2, 3 (first / last chain pieces only, no steady state) against the oracle loop: one-off check of the round-5 launch forms."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
from oracle import dctts_ref as O
W = synthetic_weights(hp, seed=1234, perturb=True)
for T in (1, 2, 3, 5):
    for B in (1, 3, 33):
        h = hp.replace(max_T=T)
        eng = Engine(W, h)
        L = synthetic_text(h, B=B, seed=7)
        Yr, _, trajr = O.synthesize(L, W, h, np.float32, run_ssrn=False)
        for graph in (0, 1):
            eng.set_decode_graph(graph)
            Y, mx = eng.text2mel(torch.from_numpy(L).cuda())
            eng.synchronize()
            err = float(np.abs(Y.cpu().numpy() - Yr).max())
            ok = (mx.cpu().numpy() == trajr).all() and err < 1e-3
            print(f"T={T} B={B} graph={graph}: max|dY| {err:.2e} trajectory {'ok' if (mx.cpu().numpy() == trajr).all() else 'DIFFERS'} {'OK' if ok else 'FAIL'}")
        del eng
