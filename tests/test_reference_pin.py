"""Pins the oracle, the product's layer tables and the text front-end to the REFERENCE'S OWN SOURCE, executed.

`/root/reference/*.py` is imported unmodified on `oracle/tf_shim.py` (a numpy stand-in for the TensorFlow symbols it touches), so
the reference's Python -- not a reading of it -- decides layer order, scope names, variable shapes, paddings, splits, the mask, the
shift and the driver loop.  TensorFlow's kernels themselves are restated by the shim: that part stays unpinned.

CPU only; skipped where /root/reference is absent (the GPU box), which consumes the fixtures these runs produced instead.
"""
import os
import tempfile

import numpy as np
import pytest

from dc_tts_amd.hyperparams import hp
from dc_tts_amd.layers import variable_shapes
from dc_tts_amd.weights import synthetic_text
from oracle import dctts_ref as O
from oracle import run_reference as RR

pytestmark = pytest.mark.skipif(not RR.available(), reason="/root/reference is not present on this machine")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_requests_exactly_the_variables_of_the_layer_tables(weights):
    """the names and shapes `Graph(mode="synthesize")` asks TensorFlow for == dc_tts_amd.layers.variable_shapes
    (+ gs/global_step, train.py:79-80), and the two scoped Savers of synthesize.py:32-40 restore every one of them."""
    with RR.reference(np.float32) as ref:
        g, sess = RR.build_synthesis_graph(ref, weights)
        req = RR.requested_variables(ref)
        restored = {n for _, n in ref.tf.RESTORED}
        assert ref.hp.max_N == hp.max_N and ref.hp.max_T == hp.max_T and ref.hp.vocab == hp.vocab
        for k in ("n_mels", "n_fft", "r", "e", "d", "c", "attention_win_size", "sr", "hop_length", "win_length", "power", "n_iter",
                  "preemphasis", "max_db", "ref_db", "dropout_rate", "B", "lr"):
            assert getattr(ref.hp, k) == getattr(hp, k), k
    spec = {n: tuple(s) for n, s in variable_shapes(hp).items()}
    assert req.pop("gs/global_step") == ()
    assert list(req) == list(spec) or set(req) == set(spec)
    assert req == spec
    assert restored == set(spec) | {"gs/global_step"}


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-5)])
def test_oracle_equals_the_reference_graph(weights, dtype, tol):
    """Every tensor of train.py:48-80's synthesize graph, computed by the reference's networks.py / modules.py, against
    oracle/dctts_ref.py on the same feeds (float64: structure only; float32: TF's own layer-norm formula vs the oracle's)."""
    T = 12
    L = synthetic_text(hp, B=2, seed=11)
    mels = np.random.default_rng(0).random((2, T, hp.n_mels)).astype(dtype)
    prev = np.array([3, 176], np.int32)                       # the second window is clipped by the end of the text (networks.py:143)
    with RR.reference(dtype, max_T=T) as ref:
        g, sess = RR.build_synthesis_graph(ref, weights)
        S, K, V, Q, R, al, mx, Yl, Y = sess.run([g.S, g.K, g.V, g.Q, g.R, g.alignments, g.max_attentions, g.Y_logits, g.Y],
                                                {g.L: L, g.mels: mels, g.prev_max_attentions: prev})
        Zl, Z = sess.run([g.Z_logits, g.Z], {g.Y: Y})
    o = O.text2mel_graph(L, mels, prev, weights, hp.replace(max_T=T), dtype)
    assert np.array_equal(S[:, 0], np.zeros_like(S[:, 0])) and np.array_equal(S[:, 1:], mels[:, :-1])          # train.py:51
    for name, a in (("K", K), ("V", V), ("Q", Q), ("R", R), ("alignments", al), ("Y_logits", Yl), ("Y", Y)):
        assert a.dtype == dtype and a.shape == o[name].shape, name
        assert np.abs(a - o[name]).max() < tol * max(1.0, np.abs(o[name]).max()), name
    assert mx.dtype == np.int64 and np.array_equal(mx, o["max_attentions"])
    assert al.shape == (2, hp.max_N, T) and (al[1, :176] == 0).all() and (al[1, 179:] == 0).all()
    zl, z = O.SSRN(Y, weights, hp, dtype)
    assert Z.shape == (2, 4 * T, 1 + hp.n_fft // 2)
    assert np.abs(Zl - zl).max() < tol * max(1.0, np.abs(zl).max()) and np.abs(Z - z).max() < tol


def test_oracle_loop_equals_the_reference_synthesize(weights):
    """`synthesize.synthesize()` itself (load_data -> Graph -> restores -> loop -> SSRN, synthesize.py:21-64) on three Harvard
    sentences, against oracle.synthesize: trajectory integer-exact, Y / Z to float32 re-association."""
    with open(os.path.join(RR.REF_DIR, "harvard_sentences.txt"), encoding="utf-8") as f:
        lines = f.readlines()
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False, encoding="utf-8") as f:
        f.writelines(lines[:1] + lines[3:6])
    try:
        r = RR.run_synthesize(weights, test_data=f.name, max_T=14)
    finally:
        os.unlink(f.name)
    assert r["L"].shape == (3, hp.max_N) and r["L"].dtype == np.int32
    assert np.array_equal(r["L"], O.load_sentences(lines[3:6], hp))
    Y, Z, traj = O.synthesize(r["L"], weights, hp.replace(max_T=14), np.float32)
    assert np.array_equal(traj, r["traj"])
    assert np.abs(Y - r["Y"]).max() < 2e-5 and np.abs(Z - r["Z"]).max() < 2e-5
    assert r["alignments_last"].shape == (3, hp.max_N, 14)


def test_committed_fixtures_are_what_the_reference_produces(weights):
    """tests/golden/networks_seed1234.npz is regenerated here from the reference run and must equal the committed file bit for
    bit (numpy on this container's CPU is deterministic); the long loop fixtures are spot-checked by their first frames."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgr", os.path.join(GOLD, "make_golden_from_reference.py"))
    mgr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgr)
    new = mgr.networks_fixture(weights)
    old = np.load(os.path.join(GOLD, "networks_seed1234.npz"))
    assert set(old.files) == set(new)
    for k in new:
        assert np.array_equal(np.asarray(new[k]), old[k]), k
    g = np.load(os.path.join(GOLD, "harvard20_ref.npz"))
    assert g["L"].shape == (20, hp.max_N) and g["Y"].shape == (20, hp.max_T, hp.n_mels) and int(g["n_restored"]) == 290
    r = RR.run_synthesize(weights, max_T=2)                   # the first two frames do not depend on max_T (causal stack, window from 0)
    assert np.array_equal(r["L"], g["L"])
    assert np.array_equal(r["traj"], g["traj"][:, :2])
    assert np.abs(r["Y"] - g["Y"][:, :2]).max() < 1e-5              # (BLAS picks other kernels for 2 rows than for 210: fp32 re-association)


def test_text_front_end_equals_the_reference(tmp_path):
    """data_load.py:19-31,79-86 executed: load_vocab, text_normalize and load_data("synthesize") on harvard_sentences.txt and on
    awkward strings, against dc_tts_amd.data_load (the product) and the oracle's restatement."""
    from dc_tts_amd import data_load as P
    hard = ["Crème brûlée — déjà vu, naïve façade!", "  MULTIPLE    spaces\tand\ttabs ", "digits 123 & symbols #@$ stay out",
            "it's 'quoted'. really? yes.", "ÅÄÖ ñ ü ß Œ æ ø", "", "E P e p", "1. numbered like the test file", "trailing space   "]
    test_file = tmp_path / "sents.txt"
    test_file.write_text("header line\n" + "".join("%d. %s\n" % (i + 1, s) for i, s in enumerate(hard)), encoding="utf-8")
    with RR.reference(np.float32) as ref:
        c2i, i2c = ref.data_load.load_vocab()
        assert (c2i, i2c) == P.load_vocab(hp)
        with open(os.path.join(RR.REF_DIR, "harvard_sentences.txt"), encoding="utf-8") as f:
            lines = f.readlines()
        for s in hard + lines:
            assert ref.data_load.text_normalize(s) == P.text_normalize(s, hp) == O.text_normalize(s, hp.vocab), s
        ref.hp.test_data = os.path.join(RR.REF_DIR, "harvard_sentences.txt")
        Lr = ref.data_load.load_data("synthesize")
        ref.hp.test_data = str(test_file)
        Lh = ref.data_load.load_data("synthesize")
    Lp = P.load_data("synthesize", os.path.join(RR.REF_DIR, "harvard_sentences.txt"), hp)
    assert Lr.shape == (20, hp.max_N) and Lr.dtype == Lp.dtype == np.int32 and np.array_equal(Lr, Lp)
    assert np.array_equal(Lr, O.load_sentences(lines[1:], hp))
    assert np.array_equal(Lh, P.load_data("synthesize", str(test_file), hp))
    assert (Lr[np.arange(20), (Lr != 0).sum(1) - 1] == 1).all()                   # every sentence ends in E, then P


def test_vocoder_loop_equals_the_reference(weights):
    """utils.py:67-114 executed (`spectrogram2wav` -> `griffin_lim` -> `invert_spectrogram`) with librosa's three functions served by
    oracle/vocoder_ref.py: pins the loop structure (de-normalise, ** power, n_iter x (istft, stft, phase), final istft, lfilter,
    trim) of the vocoder oracle to the reference's own statement.  librosa's kernels stay a restatement."""
    from oracle import vocoder_ref as VR
    mag = np.random.default_rng(7).random((9, 1 + hp.n_fft // 2)).astype(np.float32)
    with RR.reference(np.float32, n_iter=3) as ref:
        wav_ref = ref.utils.spectrogram2wav(mag)
    wav = VR.spectrogram2wav(mag, hp.replace(n_iter=3), np.float64)
    assert wav_ref.dtype == np.float32 and wav_ref.shape == wav.shape
    assert np.abs(wav_ref - wav).max() < 1e-5 * max(1.0, np.abs(wav).max())
