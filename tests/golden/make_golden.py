"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference itself cannot run here (TensorFlow 1.x, not installable: SURVEY 8c), so these vectors are
outputs of the oracle restatement on seeded inputs with the seeded synthetic weights
(dc_tts_amd.weights.synthetic_weights(seed=1234, perturb=True)); they pin the oracle against regressions
and give the GPU parity tests a committed target.    Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dc_tts_amd.hyperparams import hp                     # noqa: E402
from dc_tts_amd.weights import synthetic_text, synthetic_weights  # noqa: E402
from oracle import dctts_ref as O                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W = synthetic_weights(hp, seed=1234, perturb=True)

# per-network vectors (small T so the fixture stays small)
rng = np.random.default_rng(1234)
L = synthetic_text(hp, B=2, seed=11)
T = 24
S = rng.random((2, T, hp.n_mels), dtype=np.float32)
K, V = O.TextEnc(L, W, hp)
Q = O.AudioEnc(S, W, hp)
prev = np.array([3, 40], np.int32)
R, al, mx = O.Attention(Q, K, V, hp.replace(max_T=T), True, prev)
lg, Y = O.AudioDec(R, W, hp)
zl, Z = O.SSRN(Y[:, :8], W, hp)
np.savez_compressed(os.path.join(HERE, "networks_seed1234.npz"), L=L, S=S, prev_max=prev,
                    K_sub=K[:, ::9, ::8], V_sub=V[:, ::9, ::8], Q_sub=Q[:, ::3, ::4], max_att=mx,
                    Y=Y, Z_sub=Z[:, :, ::16])

# config 1: Harvard sentence 1 through the restated loop (T shortened to 96 > 85 to keep CPU time small)
Lh = O.load_sentences(["1. The birch canoe slid on the smooth planks.\n"], hp)
h = hp.replace(max_T=96)
Yh, _, traj = O.synthesize(Lh, W, h, np.float32, run_ssrn=False)
np.savez_compressed(os.path.join(HERE, "config1_harvard1.npz"), L=Lh, max_T=96, Y=Yh, traj=traj)
print("golden fixtures written to", HERE)
