"""Property tests (hypothesis) of the ONE parity link that is a reading rather than a run: the per-op semantics `oracle/tf_shim.py` restates for
TensorFlow (the reference's arithmetic lives in TF) and the STFT / ISTFT `oracle/vocoder_ref.py` restates for librosa.  The reference graph
only exercises them at its own shapes; here every op is compared with an INDEPENDENT implementation -- `torch.nn.functional`, torch autograd for the
transposed convolution (TF defines `conv2d_transpose` as the gradient of the forward convolution: the gradient is taken literally), plain loops --
over random kernel sizes, dilations, lengths (shorter than the dilated kernel, odd, 1 .. 3 for the stride-2 deconvolution) and widths.  float64, so a
mismatch is structure, not rounding.  >= 200 examples per property.  CPU only; needs neither /root/reference nor a GPU."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytest.importorskip("hypothesis")      # not a declared dependency of the package: present in this image, skipped where it is not
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

from oracle import tf_shim as tf
from oracle import vocoder_ref as V

SET = dict(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
TOL = 1e-11


@pytest.fixture(autouse=True)
def _float64_shim():
    tf.set_float(np.float64)
    tf.reset_default_graph()
    yield
    tf.set_float(np.float32)
    tf.reset_default_graph()


def _run(build, feeds_values, variables):
    """Build a shim graph with `build(placeholders...)`, set the variables it created, run it."""
    tf.reset_default_graph()
    phs = [tf.placeholder(tf.float32 if v.dtype.kind == "f" else tf.int32, shape=[None] * (v.ndim - 1) + [v.shape[-1]] if v.dtype.kind == "f" else [None] * v.ndim) for v in feeds_values]
    out = build(*phs)
    g = tf.get_default_graph()
    assert set(g.variables) == set(variables), (sorted(g.variables), sorted(variables))
    for n, a in variables.items():
        assert tuple(g.variables[n].shape_) == a.shape, (n, g.variables[n].shape_, a.shape)
        g.values[n] = a
    return tf.Session().run(out, dict(zip(phs, feeds_values)))


def _rng_arrays(seed, *shapes):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(s) for s in shapes]


# ---------------------------------------------------------------- tf.layers.conv1d (modules.py:134,187)
@settings(**SET)
@given(k=st.integers(1, 4), rate=st.integers(1, 30), T=st.integers(1, 40), cin=st.integers(1, 40), cout=st.integers(1, 40), B=st.integers(1, 3),
       mode=st.sampled_from(["same", "causal"]), seed=st.integers(0, 2 ** 31))
def test_conv1d_same_and_causal(k, rate, T, cin, cout, B, mode, seed):
    """'same': (k - 1) * rate zeros in total, the smaller half on the left (tf.nn.convolution's SAME rule); the reference's CAUSAL = its own left pad of
    (k - 1) * rate zeros + 'valid' (modules.py:121-125).  Against torch conv1d (cross-correlation, explicit padding), incl. T < dilated kernel."""
    x, w, b = _rng_arrays(seed, (B, T, cin), (k, cin, cout), (cout,))
    total = (k - 1) * rate

    def build(ph):
        if mode == "causal":
            ph = tf.pad(ph, [[0, 0], [total, 0], [0, 0]])                                # modules.py:123
        return tf.layers.conv1d(ph, filters=cout, kernel_size=k, dilation_rate=rate, padding="valid" if mode == "causal" else "same", use_bias=True)
    y = _run(build, [x], {"conv1d/kernel": w, "conv1d/bias": b})
    left = total if mode == "causal" else total // 2
    xt = F.pad(torch.from_numpy(x).transpose(1, 2), (left, total - left))
    yt = F.conv1d(xt, torch.from_numpy(w).permute(2, 1, 0).contiguous(), torch.from_numpy(b), dilation=rate).transpose(1, 2).numpy()
    assert y.shape == yt.shape == (B, T, cout)
    assert np.abs(y - yt).max() <= TOL * max(1.0, np.abs(yt).max())
    if mode == "causal" and T > 1:                                                        # causality, bit for bit: a change at t0 leaves rows < t0 alone
        t0 = T // 2
        x2 = x.copy(); x2[:, t0:] += 1.0
        assert np.array_equal(_run(build, [x2], {"conv1d/kernel": w, "conv1d/bias": b})[:, :t0], y[:, :t0])


def test_conv1d_valid_shorter_than_the_kernel_raises():
    x, w, b = _rng_arrays(0, (1, 4, 3), (3, 3, 2), (2,))
    with pytest.raises(ValueError):
        _run(lambda ph: tf.layers.conv1d(ph, filters=2, kernel_size=3, dilation_rate=2, padding="valid"), [x], {"conv1d/kernel": w, "conv1d/bias": b})


# ---------------------------------------------------------------- tf.layers.conv2d_transpose (modules.py:232-239)
def _tf_same_pads(n_in, k, s):
    """TensorFlow's documented SAME rule for a FORWARD strided convolution over n_in samples: out = ceil(n_in / s), total = max((out - 1) s + k - n_in, 0),
    the smaller half in front."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return out, total // 2, total - total // 2


@settings(**SET)
@given(kw=st.integers(1, 4), s=st.integers(1, 3), W=st.integers(1, 9), cin=st.integers(1, 12), cout=st.integers(1, 12), B=st.integers(1, 2),
       seed=st.integers(0, 2 ** 31))
def test_conv2d_transpose_is_the_gradient_of_the_forward_same_convolution(kw, s, W, cin, cout, B, seed):
    """conv2d_transpose(x; kernel (1, kw, Cout, Cin), strides (1, s), 'same') := d/dy <conv2d_SAME(y; kernel, stride s), x>, y of width W s.  The right-hand
    side is taken with torch autograd over an explicitly SAME-padded torch conv2d (incl. W = 1, 2, 3 at stride 2 -- the reference's case, modules.py:237)."""
    x, w, b = _rng_arrays(seed, (B, 1, W, cin), (1, kw, cout, cin), (cout,))
    y = _run(lambda ph: tf.layers.conv2d_transpose(ph, filters=cout, kernel_size=(1, kw), strides=(1, s), padding="same"),
             [x.reshape(B, 1, W, cin)], {"conv2d_transpose/kernel": w, "conv2d_transpose/bias": b})
    Wo = W * s
    out, pl, pr = _tf_same_pads(Wo, kw, s)
    assert out == W
    yv = torch.zeros(B, cout, 1, Wo, dtype=torch.float64, requires_grad=True)
    fwd = F.conv2d(F.pad(yv, (pl, pr)), torch.from_numpy(w).permute(3, 2, 0, 1).contiguous(), stride=(1, s))      # HWIO (1, kw, I = Cout, O = Cin) -> OIHW
    assert fwd.shape == (B, cin, 1, W)
    (grad,) = torch.autograd.grad((fwd * torch.from_numpy(x).permute(0, 3, 1, 2)).sum(), yv)
    want = grad.permute(0, 2, 3, 1).numpy() + b
    assert y.shape == want.shape == (B, 1, Wo, cout)
    assert np.abs(y - want).max() <= TOL * max(1.0, np.abs(want).max())


def test_conv2d_transpose_known_phase():
    """SURVEY B.5's known answer: x = [1, 1], W = [1, 1, 1], stride 2 -> [1, 1, 2, 1] (the shifted mapping gives [1, 2, 1, 1])."""
    y = _run(lambda ph: tf.layers.conv2d_transpose(ph, filters=1, kernel_size=(1, 3), strides=(1, 2), padding="same"),
             [np.ones((1, 1, 2, 1))], {"conv2d_transpose/kernel": np.ones((1, 3, 1, 1)), "conv2d_transpose/bias": np.zeros(1)})
    assert y.reshape(-1).tolist() == [1, 1, 2, 1]


# ---------------------------------------------------------------- tf.contrib.layers.layer_norm (modules.py:60-63)
@settings(**SET)
@given(C=st.integers(2, 40), T=st.integers(1, 6), B=st.integers(1, 3), scale=st.floats(1e-3, 1e3), seed=st.integers(0, 2 ** 31))
def test_layer_norm(C, T, B, scale, seed):
    x, ga, be = _rng_arrays(seed, (B, T, C), (C,), (C,))
    x = x * scale
    y = _run(lambda ph: tf.contrib.layers.layer_norm(ph, begin_norm_axis=-1, scope="normalize"), [x], {"normalize/beta": be, "normalize/gamma": ga})
    yt = F.layer_norm(torch.from_numpy(x), (C,), torch.from_numpy(ga), torch.from_numpy(be), eps=1e-12).numpy()
    # TF evaluates x * inv + (beta - mean * inv): the two products cancel to rounding of their own size, |x| / std
    cond = np.abs(x).max(-1, keepdims=True) / np.sqrt(x.var(-1, keepdims=True) + 1e-12)
    assert (np.abs(y - yt) <= 1e-14 * (1.0 + cond) * (1.0 + np.abs(ga)) + 1e-13).all()


def test_layer_norm_known_answers():
    """row [1, 2, 3, 4], gamma 1, beta 0 -> [-1.3416408, -0.4472136, 0.4472136, 1.3416408] (biased variance); an all-equal row -> exactly 0, not NaN
    (eps = 1e-12 inside the root keeps the reciprocal finite, and x * inv - mean * inv cancels exactly)."""
    x = np.array([[[1.0, 2.0, 3.0, 4.0], [7.5, 7.5, 7.5, 7.5]]])
    y = _run(lambda ph: tf.contrib.layers.layer_norm(ph, begin_norm_axis=-1, scope="normalize"), [x], {"normalize/beta": np.zeros(4), "normalize/gamma": np.ones(4)})
    assert np.abs(y[0, 0] - np.array([-1.3416407865, -0.4472135955, 0.4472135955, 1.3416407865])).max() < 1e-9
    assert np.array_equal(y[0, 1], np.zeros(4))


# ---------------------------------------------------------------- softmax / argmax / sequence_mask / where (networks.py:140-149)
@settings(**SET)
@given(N=st.integers(1, 30), rows=st.integers(1, 5), big=st.booleans(), seed=st.integers(0, 2 ** 31))
def test_softmax_and_first_index_argmax(N, rows, big, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((rows, N))
    if big:
        a[rng.random((rows, N)) < 0.5] = float(-2 ** 32 + 1)                            # the reference's mask constant (networks.py:146)
    if N > 2:
        a[:, 1] = a[:, 0]                                                               # ties
    sm, am = tf.Session().run([tf.nn.softmax(tf.convert_to_tensor(a)), tf.argmax(tf.convert_to_tensor(a), -1)])
    smt = torch.softmax(torch.from_numpy(a), -1).numpy()
    assert np.abs(sm - smt).max() < 1e-15 and np.abs(sm.sum(-1) - 1).max() < 1e-12
    assert am.dtype == np.int64
    for r in range(rows):
        assert am[r] == min(i for i in range(N) if a[r, i] == a[r].max())                # first index on ties (TF's kernel)


@settings(**SET)
@given(lengths=st.lists(st.integers(-5, 12), min_size=1, max_size=6), maxlen=st.integers(1, 10))
def test_sequence_mask(lengths, maxlen):
    m = tf.Session().run(tf.sequence_mask(tf.convert_to_tensor(np.asarray(lengths, np.int32)), maxlen))
    assert m.dtype == np.bool_ and m.shape == (len(lengths), maxlen)
    for i, ln in enumerate(lengths):
        assert m[i].tolist() == [j < ln for j in range(maxlen)]                          # a negative length is an all-False row


def test_where_wants_equal_shapes():
    c = tf.convert_to_tensor(np.ones((2, 3), bool)); a = tf.convert_to_tensor(np.zeros((2, 3))); b = tf.convert_to_tensor(np.zeros((3,)))
    with pytest.raises(ValueError):
        tf.Session().run(tf.where(c, a, b))                                              # TF 1.x: no broadcasting in tf.where(cond, x, y)


# ---------------------------------------------------------------- tf.nn.sigmoid_cross_entropy_with_logits, tf.pad(constant_values) (train.py:90-93)
@settings(**SET)
@given(n=st.integers(1, 50), scale=st.floats(0.1, 200.0), seed=st.integers(0, 2 ** 31))
def test_sigmoid_cross_entropy(n, scale, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n) * scale; z = rng.random(n)
    got = tf.Session().run(tf.nn.sigmoid_cross_entropy_with_logits(logits=tf.convert_to_tensor(x), labels=tf.convert_to_tensor(z)))
    want = F.binary_cross_entropy_with_logits(torch.from_numpy(x), torch.from_numpy(z), reduction="none").numpy()
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


@settings(**SET)
@given(shape=st.tuples(st.integers(1, 4), st.integers(1, 5), st.integers(1, 5)), pads=st.tuples(st.integers(0, 3), st.integers(0, 3), st.integers(0, 3), st.integers(0, 3)),
       value=st.floats(-3, 3), seed=st.integers(0, 2 ** 31))
def test_pad_with_constant(shape, pads, value, seed):
    x = np.random.default_rng(seed).standard_normal(shape)
    got = tf.Session().run(tf.pad(tf.convert_to_tensor(x), [(0, 0), (pads[0], pads[1]), (pads[2], pads[3])], mode="CONSTANT", constant_values=value))
    want = F.pad(torch.from_numpy(x), (pads[2], pads[3], pads[0], pads[1]), value=value).numpy()
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- librosa.stft / istft as oracle/vocoder_ref.py restates them (utils.py:101,108-114)
@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(n_fft=st.sampled_from([16, 32, 64, 128]), win_frac=st.floats(0.3, 1.0), hop_div=st.integers(2, 8), frames=st.integers(2, 12), extra=st.integers(0, 7),
       seed=st.integers(0, 2 ** 31))
def test_stft_and_istft_against_torch(n_fft, win_frac, hop_div, frames, extra, seed):
    """Random (n_fft, win_length <= n_fft, hop) and signal lengths that are NOT multiples of the hop: centred reflect padding, centre-padded periodic Hann,
    window-sum-square normalisation -- against torch.stft / torch.istft (an independent implementation of the same conventions)."""
    win = max(2, int(n_fft * win_frac)); hop = max(1, win // hop_div)
    h = types.SimpleNamespace(n_fft=n_fft, win_length=win, hop_length=hop)
    n = hop * (frames - 1) + extra
    if n <= n_fft // 2:                                                                  # reflect padding needs more than n_fft // 2 samples (librosa raises too)
        n = n_fft // 2 + 1 + extra
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(n)
    w = torch.from_numpy(V.hann_periodic(win, np.float64))
    St = torch.stft(torch.from_numpy(y), n_fft, hop, win, window=w, center=True, pad_mode="reflect", return_complex=True).numpy()
    S = V.stft(y, h, np.float64)
    assert S.shape == St.shape == (1 + n_fft // 2, 1 + n // hop)
    assert np.abs(S - St).max() < 1e-10
    X = rng.standard_normal(S.shape) + 1j * rng.standard_normal(S.shape)
    X[0].imag = 0; X[-1].imag = 0                                                        # a real signal's DC / Nyquist bins
    nf = S.shape[1]
    wss = V.window_sumsquare(h, nf, np.float64)[n_fft // 2: n_fft // 2 + hop * (nf - 1)]
    if wss.size and wss.min() > 1e-8:                                                    # torch.istft refuses windows whose overlap-add has (near) zeros
        yt = torch.istft(torch.from_numpy(X), n_fft, hop, win, window=w, center=True).numpy()
        yo = V.istft(X, h, np.float64)
        assert yo.shape == yt.shape == (hop * (nf - 1),)
        assert np.abs(yo - yt).max() < 1e-9 * max(1.0, np.abs(yt).max())
