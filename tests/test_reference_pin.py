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


# ---------------------------------------------------------------- the training graph of train.py:82-131 (round 5: pins oracle/train_ref.py)
def _train_batch(h, B, N, T, seed):
    rng = np.random.default_rng(seed)
    L = np.zeros((B, N), np.int32)
    for b in range(B):
        n = int(rng.integers(N // 2, N + 1))
        L[b, :n - 1] = rng.integers(2, len(h.vocab), n - 1); L[b, n - 1] = 1
    mels = rng.random((B, T, h.n_mels))
    mags = rng.random((B, h.r * T, h.n_linear))
    return L, mels, mags


def _w64(weights):
    return {k: np.asarray(v, np.float64) for k, v in weights.items()}


@pytest.mark.parametrize("gs", [0, 7000])
def test_training_graph_text2mel_losses_equal_the_oracle(weights, gs):
    """`Graph(num=1)` (mode="train") built by the reference's own constructor on the shim -- only `get_batch()` replaced by placeholders -- against
    oracle/train_ref.py in float64: Y, alignments, the three losses of train.py:85-100 incl. the -1 padding / crop rule of :93 at N < max_N and T < max_T,
    mask_sum, the guided-attention constant (utils.py:134-140) and the Noam learning rate (utils.py:142-145).  Dropout off (rate 0): TF's random stream
    cannot be reproduced; the next test pins WHERE dropout sits."""
    from oracle import train_ref as TR
    B, N, T = 2, 40, 24
    L, mels, mags = _train_batch(hp, B, N, T, 3)
    with RR.reference(np.float64, B=B, dropout_rate=0.0) as ref:
        g, sess, (pL, pmels, pmags) = RR.build_training_graph(ref, 1, {k: v for k, v in weights.items() if k.startswith("Text2Mel/")}, global_step=gs)
        assert set(RR.requested_variables(ref)) == {k for k in weights if k.startswith("Text2Mel/")} | {"gs/global_step"}
        Y, al, lm, lb, la, loss, msum, lr, gts = sess.run([g.Y, g.alignments, g.loss_mels, g.loss_bd1, g.loss_att, g.loss, g.mask_sum, g.lr, g.gts], {pL: L, pmels: mels})
        assert len(g.gvs) == len(g.clipped) == 209                                     # one (gradient, variable) pair per Text2Mel variable (train.py:119-124)
        with pytest.raises(NotImplementedError):
            sess.run(g.train_op, {pL: L, pmels: mels})                                  # TensorFlow's autodiff / Adam are NOT restated by the shim
        assert ("scalar", "train/loss_att") in ref.tf.SUMMARIES and ("scalar", "lr") in ref.tf.SUMMARIES
        ga = ref.utils.guided_attention()
    W = _w64(weights)
    losses, grads = TR.train_grads(1, W, gs, (L, mels), hp)
    assert abs(lm - losses[0]) < 1e-12 and abs(lb - losses[1]) < 1e-12 and abs(la - losses[2]) < 1e-12 and abs(loss - sum(losses)) < 1e-12
    assert msum == B * N * T                                                            # only the real (N, T) block of the padded alignments counts (train.py:93-96)
    assert abs(lr - TR.learning_rate_decay(hp.lr, gs)) < 1e-18
    assert ga.dtype == np.float32 and ga.shape == (hp.max_N, hp.max_T)
    assert np.array_equal(ga.astype(np.float64), TR.guided_attention(hp.max_N, hp.max_T)) and np.array_equal(np.asarray(gts), ga)
    assert set(grads) == {k for k in weights if k.startswith("Text2Mel/")}


def test_training_graph_ssrn_losses_equal_the_oracle(weights):
    """`Graph(num=2)`: SSRN on the ground-truth mels (train.py:69-72), loss_mags + loss_bd2 (train.py:102-110), float64."""
    from oracle import train_ref as TR
    B, T = 2, 6
    _, mels, mags = _train_batch(hp, B, 8, T, 4)
    with RR.reference(np.float64, B=B, dropout_rate=0.0) as ref:
        g, sess, (pL, pmels, pmags) = RR.build_training_graph(ref, 2, {k: v for k, v in weights.items() if k.startswith("SSRN/")})
        assert set(RR.requested_variables(ref)) == {k for k in weights if k.startswith("SSRN/")} | {"gs/global_step"}
        Z, l1, l2, loss = sess.run([g.Z, g.loss_mags, g.loss_bd2, g.loss], {pmels: mels, pmags: mags})
        assert len(g.gvs) == 80 and not hasattr(g, "loss_att")
    losses, grads = TR.train_grads(2, _w64(weights), 0, (mels, mags), hp)
    assert abs(l1 - losses[0]) < 1e-12 and abs(l2 - losses[1]) < 1e-12 and abs(loss - sum(losses)) < 1e-12
    assert Z.shape == (B, 4 * T, hp.n_linear)


def test_training_graph_dropout_sits_where_the_oracle_puts_it(weights):
    """training=True puts `tf.layers.dropout(rate=hp.dropout_rate)` behind every conv1d / hc / conv1d_transpose block and nowhere else (modules.py:139,195,245).
    The shim's dropout calls -- made by the reference's modules.py -- are given the ORACLE's counter-hash mask for (network, layer index): the losses of both
    graphs then agree to rounding, which they cannot if one block's dropout were missing, doubled, at another rate or keyed to another layer."""
    from oracle import train_ref as TR
    B, N, T = 2, 30, 16
    L, mels, mags = _train_batch(hp, B, N, T, 5)
    seed, gs = 17, 3
    lists = {"Text2Mel/TextEnc": TR.TEXTENC_LAYERS, "Text2Mel/AudioEnc": TR.AUDIOENC_LAYERS, "Text2Mel/AudioDec": TR.AUDIODEC_LAYERS, "SSRN": TR.SSRN_LAYERS}

    def hook(scope, rate):
        prefix, layer = scope.rsplit("/", 1)
        li = [i for i, l in enumerate(lists[prefix]) if l.scope == layer]
        assert len(li) == 1 and rate == hp.dropout_rate, (scope, rate)
        key = TR.layer_key(seed, gs, prefix, li[0])
        return lambda x: TR.dropout(x, key, rate)
    for num, feed_names in ((1, "Lm"), (2, "mg")):
        with RR.reference(np.float64, B=B) as ref:
            ref.tf.DROPOUT_HOOK = hook
            try:
                scope = "Text2Mel/" if num == 1 else "SSRN/"
                g, sess, (pL, pmels, pmags) = RR.build_training_graph(ref, num, {k: v for k, v in weights.items() if k.startswith(scope)}, global_step=gs)
                calls = list(ref.tf.DROPOUT_CALLS)
                if num == 1:
                    got = sess.run([g.loss_mels, g.loss_bd1, g.loss_att], {pL: L, pmels: mels})
                else:
                    got = sess.run([g.loss_mags, g.loss_bd2], {pmels: mels[:, :4], pmags: mags[:, :16]})
            finally:
                ref.tf.DROPOUT_HOOK = None
        n_blocks = (len(TR.TEXTENC_LAYERS) - 1 + len(TR.AUDIOENC_LAYERS) + len(TR.AUDIODEC_LAYERS)) if num == 1 else len(TR.SSRN_LAYERS)
        assert len(calls) == n_blocks and all(tr and r == hp.dropout_rate for _, r, tr in calls)
        batch = (L, mels) if num == 1 else (mels[:, :4], mags[:, :16])
        want, _ = TR.train_grads(num, _w64(weights), gs, batch, hp, dropout_seed=seed)
        for a, b in zip(got, want):
            assert abs(a - b) < 1e-12, (num, got, want)


def test_oracle_gradients_are_the_derivatives_of_the_reference_graphs_loss():
    """`optimizer.compute_gradients(self.loss)` (train.py:119) is TensorFlow's autodiff of the loss the reference BUILT.  oracle/train_ref.py writes every
    derivative out by hand; here each is compared with a central finite difference of the REFERENCE GRAPH's own loss (its networks.py / modules.py / train.py
    executed on the shim, float64) -- on a narrow model (e = d = 8, c = 12, 6 mel bins, 9 linear bins: every layer of all four networks is still there), one
    randomly chosen entry of EVERY variable, Text2Mel and SSRN."""
    from dc_tts_amd.weights import synthetic_weights
    from oracle import train_ref as TR
    h = hp.replace(e=8, d=8, c=12, n_mels=6, n_fft=16, max_N=14, max_T=11, B=2, dropout_rate=0.0)
    Wf = synthetic_weights(h, seed=5, perturb=True)
    W = _w64(Wf)
    L, mels, mags = _train_batch(h, 2, 10, 7, 6)
    rng = np.random.default_rng(8)
    over = dict(e=h.e, d=h.d, c=h.c, n_mels=h.n_mels, n_fft=h.n_fft, max_N=h.max_N, max_T=h.max_T, B=2, dropout_rate=0.0)
    eps = 1e-6
    for num in (1, 2):
        scope = "Text2Mel/" if num == 1 else "SSRN/"
        batch = (L, mels) if num == 1 else (mels, mags)
        _, grads = TR.train_grads(num, W, 0, batch, h)
        names = [k for k in W if k.startswith(scope)]
        assert set(grads) == set(names)
        with RR.reference(np.float64, **over) as ref:
            g, sess, (pL, pmels, pmags) = RR.build_training_graph(ref, num, {k: W[k] for k in names})
            feeds = {pL: L, pmels: mels} if num == 1 else {pmels: mels, pmags: mags}
            vals = ref.tf.get_default_graph().values
            worst = 0.0
            for n in names:
                idx = tuple(int(rng.integers(0, s)) for s in W[n].shape)
                if n.endswith("lookup_table") and idx[0] == 0:
                    idx = (int(L[0, 0]),) + idx[1:]                  # row 0 of the table is replaced by zeros at lookup time (modules.py:36-38): no gradient there
                keep = vals[n][idx]
                vals[n][idx] = keep + eps; lp = float(sess.run(g.loss, feeds))
                vals[n][idx] = keep - eps; lm = float(sess.run(g.loss, feeds))
                vals[n][idx] = keep
                fd = (lp - lm) / (2 * eps)
                an = float(grads[n][idx])
                err = abs(fd - an) / max(1e-4, abs(an), abs(fd))
                worst = max(worst, err)
                assert err < 2e-5, (n, idx, fd, an)
        assert worst < 2e-5
