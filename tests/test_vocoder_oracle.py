"""CPU tests of oracle/vocoder_ref.py (the restatement of utils.py:67-114).  librosa is absent (parity unpinned), so the
restatement is pinned to what IS here: scipy (window, lfilter), torch.stft / torch.istft (an independent implementation of
the same librosa conventions), closed-form known answers, and the committed golden fixture."""
import os

import numpy as np
import pytest
import scipy.signal
import torch

from dc_tts_amd.hyperparams import hp
from oracle import vocoder_ref as V

HERE = os.path.dirname(os.path.abspath(__file__))


def test_window_matches_scipy():
    w = scipy.signal.get_window("hann", hp.win_length, fftbins=True)
    assert np.abs(V.hann_periodic(hp.win_length, np.float64) - w).max() < 1e-15
    pw = V.padded_window(hp, np.float64)
    lpad = (hp.n_fft - hp.win_length) // 2
    assert lpad == 473 and pw[:lpad].max() == 0 and pw[lpad + hp.win_length:].max() == 0
    assert np.abs(pw[lpad:lpad + hp.win_length] - w).max() < 1e-15


def test_deemphasis_matches_scipy_lfilter():
    x = np.random.default_rng(0).standard_normal(5000)
    ref = scipy.signal.lfilter([1], [1, -hp.preemphasis], x)          # utils.py:89
    assert np.abs(V.deemphasis(x, hp, np.float64) - ref).max() < 1e-12
    assert np.abs(V.deemphasis(x, hp, np.float32) - ref).max() < 1e-4


def test_stft_known_answer_cosine():
    """A cosine exactly on bin k: |X[k]| = sum(window)/2 in every interior frame, other far bins ~ 0."""
    k = 100
    n = hp.hop_length * 39
    y = np.cos(2 * np.pi * k * np.arange(n) / hp.n_fft)
    S = V.stft(y, hp, np.float64)
    assert S.shape == (hp.n_linear, 40)
    w = V.padded_window(hp, np.float64)
    assert np.abs(np.abs(S[k, 5:-5]) - w.sum() / 2).max() < 1e-6 * w.sum()      # up to the image at -k's sidelobe
    assert np.abs(S[k + 20, 5:-5]).max() < 1e-3 * w.sum() / 2                     # Hann sidelobe ~11 window-bins away


def test_stft_istft_match_torch():
    rng = np.random.default_rng(1)
    n = hp.hop_length * 29
    y = rng.standard_normal(n)
    win = torch.from_numpy(V.hann_periodic(hp.win_length, np.float64))
    St = torch.stft(torch.from_numpy(y), hp.n_fft, hp.hop_length, hp.win_length, window=win, center=True,
                    pad_mode="reflect", return_complex=True).numpy()
    S = V.stft(y, hp, np.float64)
    assert S.shape == St.shape == (hp.n_linear, 30)
    assert np.abs(S - St).max() < 1e-9
    # istft of an arbitrary (inconsistent) spectrogram
    X = rng.standard_normal((hp.n_linear, 30)) + 1j * rng.standard_normal((hp.n_linear, 30))
    yt = torch.istft(torch.from_numpy(X), hp.n_fft, hp.hop_length, hp.win_length, window=win, center=True).numpy()
    yo = V.istft(X, hp, np.float64)
    assert yo.shape == yt.shape == (n,)
    assert np.abs(yo - yt).max() < 1e-9
    # perfect reconstruction of a consistent one
    assert np.abs(V.istft(S, hp, np.float64) - y).max() < 1e-9


def test_float32_mode_close_to_float64():
    rng = np.random.default_rng(2)
    Z = rng.random((40, hp.n_linear)).astype(np.float32)
    a = V.griffin_lim(V.denormalize(Z, hp, np.float32), hp, np.float32, 5)
    b = V.griffin_lim(V.denormalize(Z, hp, np.float64), hp, np.float64, 5)
    assert a.dtype == np.float32 and np.abs(a - b).max() < 1e-4 * np.abs(b).max()


def test_denormalize_known_answer():
    m = np.array([[0.0, 0.5, 1.0, 1.5, -1.0]], np.float32)            # clip -> 0, .5, 1, 1, 0
    d = V.denormalize(m, hp, np.float64)[:, 0]
    exp = (10.0 ** ((np.array([0, .5, 1, 1, 0]) * 100 - 100 + 20) * 0.05)) ** 1.5
    assert np.allclose(d, exp, rtol=1e-12)


def test_griffin_lim_reduces_inconsistency():
    """Griffin-Lim's defining property: | |stft(y)| - spec | shrinks with iterations."""
    rng = np.random.default_rng(3)
    n = hp.hop_length * 39
    t = np.arange(n) / hp.sr
    sig = np.sin(2 * np.pi * (300 + 400 * t) * t)
    spec = np.abs(V.stft(sig, hp, np.float64))
    err = []
    for it in (0, 5, 30):
        y = V.griffin_lim(spec, hp, np.float64, it)
        err.append(np.linalg.norm(np.abs(V.stft(y, hp, np.float64)) - spec) / np.linalg.norm(spec))
    assert err[0] > err[1] > err[2]


def test_trim_bounds_known_answer():
    n = 512 * 40
    y = np.zeros(n)
    y[512 * 10:512 * 20] = np.sin(np.arange(512 * 10) * 0.3)
    s, e = V.trim_bounds(y)
    # frame j covers [512 j - 1024, 512 j + 1024): the first one touching the burst [5120, 10240) is j = 9, the last j = 21
    assert (s, e) == (512 * 9, 512 * 22)
    assert V.trim_bounds(np.zeros(n)) == (0, min(n, (n // 512 + 1) * 512))     # all frames equal the max: nothing is trimmed
    z = y.copy(); z[:] += 1e-9
    assert V.trim_bounds(z) == (512 * 9, 512 * 22)


def test_golden_vocoder_fixture():
    g = np.load(os.path.join(HERE, "golden", "vocoder_seed7.npz"))
    wav, (s, e) = V.spectrogram2wav(g["mag"], hp, np.float32, n_iter=int(g["n_iter"]), return_untrimmed=True)
    assert (s, e) == tuple(int(v) for v in g["bounds"])
    assert np.abs(wav - g["wav"]).max() <= 1e-6 * np.abs(g["wav"]).max()
