"""Runs the reference's OWN Python (`/root/reference/*.py`, unmodified, imported from where it lies) on top of `oracle/tf_shim.py`
--  TEST INFRASTRUCTURE, NOT PRODUCT.  Used by tests/test_reference_pin.py and tests/golden/make_golden_from_reference.py.

What this pins: the layer lists, variable names and shapes, paddings, splits, the attention mask, the decoder-input shift
(`train.py:51`), the driver loop (`synthesize.py:45-57`), the text front-end (`data_load.py:19-31,79-86`) and the vocoder's
loop structure (`utils.py:67-114`) are the reference's own statements, executed.  What it does not pin: TensorFlow's and
librosa's kernels, which `tf_shim.py` / `vocoder_ref.py` restate.  `/root/reference` exists only in the build
container: everything here is skipped on the GPU box, which consumes the committed fixtures instead.
"""
import contextlib
import importlib
import os
import sys
import tempfile
import types

import numpy as np

from . import tf_shim
from . import vocoder_ref

REF_DIR = os.environ.get("DCTTS_REFERENCE_DIR", "/root/reference")
REF_MODULES = ("hyperparams", "modules", "networks", "utils", "data_load", "train", "synthesize")


def available():
    return os.path.isfile(os.path.join(REF_DIR, "networks.py"))


def _librosa_stub(hp):
    """`librosa` as `utils.py:67-114` uses it, served by the restated librosa-0.6 algorithms of oracle/vocoder_ref.py (float64)."""
    class _HP:
        n_fft, hop_length, win_length = hp.n_fft, hp.hop_length, hp.win_length

    lib = types.ModuleType("librosa")

    def stft(y, n_fft=2048, hop_length=None, win_length=None, **k):
        assert (n_fft, hop_length, win_length) == (_HP.n_fft, _HP.hop_length, _HP.win_length)
        return vocoder_ref.stft(np.asarray(y), _HP, np.float64)

    def istft(stft_matrix, hop_length=None, win_length=None, window="hann", **k):
        assert (hop_length, win_length, window) == (_HP.hop_length, _HP.win_length, "hann")
        return vocoder_ref.istft(np.asarray(stft_matrix), _HP, np.float64)

    def trim(y, top_db=60, frame_length=2048, hop_length=512):
        s, e = vocoder_ref.trim_bounds(np.asarray(y), top_db, frame_length, hop_length, np.float64)
        return y[s:e], np.array([s, e])

    lib.stft, lib.istft = stft, istft
    lib.effects = types.ModuleType("librosa.effects")
    lib.effects.trim = trim
    lib.filters = types.ModuleType("librosa.filters")
    lib.load = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("feature extraction is out of scope"))
    return lib


@contextlib.contextmanager
def reference(float_dtype=np.float32, **hp_overrides):
    """Context: the reference's modules freshly imported on the shim.  Yields a namespace with one attribute per module
    (`ref.networks`, `ref.train`, ...), `ref.hp` (the reference's Hyperparams class, patched with `hp_overrides` BEFORE the
    modules that read it at import time are loaded) and `ref.tf` (the shim).  sys.modules / sys.path are restored on exit."""
    if not available():
        raise FileNotFoundError(REF_DIR)
    saved = {n: sys.modules.get(n) for n in REF_MODULES + ("tensorflow", "librosa")}
    for n in REF_MODULES:
        sys.modules.pop(n, None)
    tf_shim.set_float(float_dtype)
    tf_shim.reset_default_graph()
    tf_shim.CHECKPOINTS.clear()
    del tf_shim.RESTORED[:]
    stubs = [n for n in tf_shim.install() if n.startswith("matplotlib")]      # only when matplotlib is not installed
    sys.path.insert(0, REF_DIR)
    old_write = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                     # /root/reference is read-only
    try:
        hyper = importlib.import_module("hyperparams")
        for k, v in hp_overrides.items():
            if not hasattr(hyper.Hyperparams, k):
                raise AttributeError(k)
            setattr(hyper.Hyperparams, k, v)
        sys.modules["librosa"] = _librosa_stub(hyper.Hyperparams)
        ns = types.SimpleNamespace(hp=hyper.Hyperparams, tf=tf_shim, hyperparams=hyper)
        for n in REF_MODULES[1:]:
            setattr(ns, n, importlib.import_module(n))
        yield ns
    finally:
        sys.dont_write_bytecode = old_write
        sys.path.remove(REF_DIR)
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
        for n in ["tensorflow.nn", "tensorflow.layers", "tensorflow.contrib", "tensorflow.contrib.layers", "tensorflow.train", "tensorflow.summary"] + stubs:
            sys.modules.pop(n, None)
        tf_shim.set_float(np.float32)


def requested_variables(ref):
    """{op name: shape} of every variable the reference's graph construction asked TensorFlow for, in creation order."""
    return {n: tuple(v.shape_) for n, v in ref.tf.get_default_graph().variables.items()}


def build_synthesis_graph(ref, weights):
    """`g = Graph(mode="synthesize")` (synthesize.py:26) + the two scoped restores of synthesize.py:32-40, from a weights dict."""
    tf = ref.tf
    g = ref.train.Graph(mode="synthesize")
    sess = tf.Session()
    sess.run(tf.global_variables_initializer())
    src = dict(weights)
    src.setdefault("gs/global_step", np.int32(0))
    tf.register_checkpoint("shim-1", src)
    tf.register_checkpoint("shim-2", src)
    tf.train.Saver(var_list=tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, "Text2Mel")).restore(sess, tf.train.latest_checkpoint("shim-1"))
    tf.train.Saver(var_list=tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES, "SSRN")
                   + tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, "gs")).restore(sess, tf.train.latest_checkpoint("shim-2"))
    return g, sess


def run_synthesize(weights, test_data=None, float_dtype=np.float32, vocoder=False, **hp_overrides):
    """Calls the reference's `synthesize.synthesize()` itself (synthesize.py:21-64): load_data("synthesize") -> Graph ->
    the two restores -> the 210-step loop -> one SSRN pass -> (optionally) spectrogram2wav + wav files in a scratch directory.
    Returns dict(L, Y, traj, Z, alignments_last, wavs, restored, variables)."""
    tmp = tempfile.mkdtemp(prefix="dctts_ref_")
    over = dict(hp_overrides)
    over.setdefault("test_data", test_data or os.path.join(REF_DIR, "harvard_sentences.txt"))
    over["sampledir"] = os.path.join(tmp, "samples")
    over["logdir"] = os.path.join(tmp, "logdir")
    with reference(float_dtype, **over) as ref:
        tf = ref.tf
        src = dict(weights)
        src.setdefault("gs/global_step", np.int32(0))
        tf.register_checkpoint(over["logdir"] + "-1", src)
        tf.register_checkpoint(over["logdir"] + "-2", src)
        wavs = []
        syn = ref.synthesize
        if not vocoder:
            syn.spectrogram2wav = lambda mag: np.zeros(8, np.float32)      # the metric's boundary: Griffin-Lim is outside
        else:
            orig = syn.spectrogram2wav
            syn.spectrogram2wav = lambda mag: (wavs.append(orig(mag)) or wavs[-1])
        syn.tqdm = lambda it, *a, **k: it
        tf_shim.LOG_RUNS = True
        try:
            syn.synthesize()
        finally:
            tf_shim.LOG_RUNS = False
        runs = [r for r in tf_shim.RUN_LOG if r["feeds"]]
        steps, last = runs[:-1], runs[-1]
        T = ref.hp.max_T
        assert len(steps) == T and len(last["feeds"]) == 1, (len(steps), T)
        Y = last["feeds"][last["feed_tensors"][0].name]                     # synthesize.py:57 feeds g.Y
        Z = last["results"][0]
        L = steps[0]["feeds"][steps[0]["feed_tensors"][0].name]
        traj = np.stack([s["results"][2][:, j] for j, s in enumerate(steps)], axis=1)     # synthesize.py:54
        out = dict(L=np.asarray(L), Y=np.asarray(Y), traj=traj.astype(np.int64), Z=np.asarray(Z),
                   alignments_last=np.asarray(steps[-1]["results"][3]), wavs=wavs,
                   restored=list(tf_shim.RESTORED), variables=requested_variables(ref))
        del tf_shim.RUN_LOG[:]
    return out


def build_training_graph(ref, num, weights, global_step=0):
    """`g = Graph(num=num)` -- i.e. mode="train", train.py:139 -- built by the reference's own constructor, with ONE substitution: the input queue
    `get_batch()` (train.py:39: TF reader threads over the LJ Speech files, out of scope) is replaced by placeholders of the same dtypes and
    static shapes (data_load.py:117-131).  Everything behind it -- the decoder-input shift, the four networks with training=True, the five losses,
    the guided-attention constant, the Noam schedule, the optimizer wiring, the summaries -- runs as written.  Variables come from `weights`
    (Text2Mel or SSRN scope, whichever Graph(num) creates) and `gs/global_step`.  Returns (g, sess, feeds) with feeds = (L, mels, mags)."""
    tf = ref.tf
    hp = ref.hp
    L = tf.placeholder(tf.int32, shape=(hp.B, None))
    mels = tf.placeholder(tf.float32, shape=(hp.B, None, hp.n_mels))
    mags = tf.placeholder(tf.float32, shape=(hp.B, None, hp.n_fft // 2 + 1))
    fnames = tf.placeholder(tf.string, shape=(hp.B,))
    ref.train.get_batch = lambda: (L, mels, mags, fnames, 1)
    g = ref.train.Graph(num=num)
    sess = tf.Session()
    sess.run(tf.global_variables_initializer())
    src = dict(weights)
    src["gs/global_step"] = np.int32(global_step)
    tf.register_checkpoint("shim-train", src)
    tf.train.Saver().restore(sess, tf.train.latest_checkpoint("shim-train"))      # every variable Graph(num) created: the Supervisor's restore (train.py:142)
    return g, sess, (L, mels, mags)
