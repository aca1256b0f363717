"""Regenerates the golden fixtures of the synthesis path by EXECUTING THE REFERENCE'S OWN PYTHON
(`/root/reference/{hyperparams,modules,networks,train,synthesize,data_load}.py`, imported unmodified) on the numpy-backed
TensorFlow stand-in `oracle/tf_shim.py` (TensorFlow itself cannot be installed here: SURVEY 8c).

  networks_seed1234.npz   Graph(mode="synthesize") (train.py:43-80): K, V, Q, max_attentions, Y, Z for seeded inputs (T = 24)
  config1_harvard1.npz    BASELINE configs[0]: synthesize.synthesize() on Harvard sentence 1, max_T = 96
  harvard20_ref.npz       the reference's literal synthesis workload: synthesize.synthesize() on all 20 Harvard sentences as one
                          batch (data_load.py:79-86, synthesize.py:23), max_T = 210, max_N = 180: L, Y, the attention trajectory,
                          the alignments fetched at the last step and a sub-sampled Z

Weights: dc_tts_amd.weights.synthetic_weights(seed=1234, perturb=True) served through the shim's Saver exactly as
synthesize.py:32-40 restores them.  Arithmetic: float32.  Runs only where /root/reference exists (the build container);
the GPU box consumes the committed files.      Usage:  python tests/golden/make_golden_from_reference.py [--skip-harvard20]
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dc_tts_amd.hyperparams import hp                     # noqa: E402
from dc_tts_amd.weights import synthetic_text, synthetic_weights  # noqa: E402
from oracle import run_reference as RR                    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
Z_SUB = (slice(None), slice(None, None, 12), slice(None, None, 8))     # harvard20: every 12th frame, every 8th bin


def networks_fixture(W):
    rng = np.random.default_rng(1234)
    L = synthetic_text(hp, B=2, seed=11)
    T = 24
    S = rng.random((2, T, hp.n_mels), dtype=np.float32)
    prev = np.array([3, 40], np.int32)
    with RR.reference(np.float32, max_T=T) as ref:
        g, sess = RR.build_synthesis_graph(ref, W)
        # g.S is fed directly: train.py:51's shift is exercised by the loop fixtures, here AudioEnc sees S as given
        K, V, Q, mx, Y = sess.run([g.K, g.V, g.Q, g.max_attentions, g.Y], {g.L: L, g.S: S, g.prev_max_attentions: prev})
        Z = sess.run(g.Z, {g.Y: Y[:, :8]})
    return dict(L=L, S=S, prev_max=prev, K_sub=K[:, ::9, ::8], V_sub=V[:, ::9, ::8], Q_sub=Q[:, ::3, ::4], max_att=mx,
                Y=Y, Z_sub=Z[:, :, ::16])


def harvard_lines():
    with open(os.path.join(RR.REF_DIR, "harvard_sentences.txt"), encoding="utf-8") as f:
        return f.readlines()


def main():
    W = synthetic_weights(hp, seed=1234, perturb=True)
    t0 = time.time()
    np.savez_compressed(os.path.join(HERE, "networks_seed1234.npz"), **networks_fixture(W))
    print("networks_seed1234.npz  %.0f s" % (time.time() - t0), flush=True)

    t0 = time.time()
    lines = harvard_lines()
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False, encoding="utf-8") as f:
        f.writelines(lines[:2])                                   # header + sentence 1
    r = RR.run_synthesize(W, test_data=f.name, max_T=96)
    os.unlink(f.name)
    np.savez_compressed(os.path.join(HERE, "config1_harvard1.npz"), L=r["L"], max_T=96, Y=r["Y"], traj=r["traj"])
    print("config1_harvard1.npz  %.0f s" % (time.time() - t0), flush=True)

    if "--skip-harvard20" in sys.argv:
        return
    t0 = time.time()
    r = RR.run_synthesize(W)                                      # hp as the reference ships it: 20 sentences, N = 180, T = 210
    al = r["alignments_last"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "harvard20_ref.npz"), L=r["L"], Y=r["Y"], traj=r["traj"], alignments_last=al,
                        Z_sub=r["Z"][Z_SUB], z_sub_step=np.array([12, 8]),
                        variables=np.array(sorted(r["variables"])), n_restored=len(r["restored"]))
    print("harvard20_ref.npz  %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
