"""CPU restatement of the reference's vocoder tail: `utils.py:67-114` (`spectrogram2wav`, `griffin_lim`,
`invert_spectrogram`), called once per utterance from `synthesize.py:61-64`.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product (dc_tts_amd/).

PARITY UNPINNED.  The arithmetic lives in third-party packages that are absent here and unpinned in the reference
(`librosa` -- "librosa" with no version in README.md:9 -- and `scipy.signal.lfilter`; scipy IS installed, so `deemphasis`
below is checked against it in tests/test_vocoder_oracle.py).  What is restated is the published algorithm of
librosa 0.6 (contemporary with the reference, Feb 2018):
  * `librosa.stft(y, n_fft, hop, win_length)`: periodic Hann of win_length zero-padded to n_fft about its centre
    (`util.pad_center`), signal reflect-padded by n_fft//2, frames every hop, rfft  -> (1 + n_fft//2, n_frames).
  * `librosa.istft(S, hop, win_length, window="hann")`: per frame irfft * window, overlap-add in frame order, divide by the
    overlap-added squared window where it exceeds `tiny`, drop n_fft//2 samples at both ends.
  * `librosa.effects.trim(y)`: top_db=60, frame_length=2048, hop_length=512, centred (reflect) RMS frames, ref = max.
librosa mixes float32 signals with float64 windows; here `dtype` selects ONE arithmetic for everything (float32 = what the
HIP path computes in; float64 to tell algorithm errors from rounding).
"""
import numpy as np


def hann_periodic(win_length, dtype=np.float32):
    """scipy.signal.get_window('hann', win_length, fftbins=True), as librosa.filters.get_window calls it."""
    n = np.arange(win_length, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(dtype)


def padded_window(hp, dtype=np.float32):
    """util.pad_center(window, n_fft): lpad = (n_fft - win_length) // 2."""
    w = np.zeros(hp.n_fft, dtype)
    lpad = (hp.n_fft - hp.win_length) // 2
    w[lpad:lpad + hp.win_length] = hann_periodic(hp.win_length, dtype)
    return w


def window_sumsquare(hp, n_frames, dtype=np.float32):
    """Overlap-added squared window over n_fft + hop*(n_frames-1) samples, accumulated in frame order (istft's divisor)."""
    w2 = padded_window(hp, dtype) ** 2
    out = np.zeros(hp.n_fft + hp.hop_length * (n_frames - 1), dtype)
    for i in range(n_frames):
        s = i * hp.hop_length
        out[s:s + hp.n_fft] = out[s:s + hp.n_fft] + w2
    return out


def istft(S, hp, dtype=np.float32):
    """librosa.istft(S, hp.hop_length, win_length=hp.win_length, window="hann")  (utils.py:108-114).  S: (1+n_fft//2, n_frames)."""
    cdtype = np.complex64 if dtype == np.float32 else np.complex128
    n_fft, hop = hp.n_fft, hp.hop_length
    n_frames = S.shape[1]
    w = padded_window(hp, dtype)
    frames = np.fft.irfft(S.astype(cdtype).T, n=n_fft, axis=1).astype(dtype) * w[None, :]      # (n_frames, n_fft)
    y = np.zeros(n_fft + hop * (n_frames - 1), dtype)
    for i in range(n_frames):
        s = i * hop
        y[s:s + n_fft] = y[s:s + n_fft] + frames[i]
    wss = window_sumsquare(hp, n_frames, dtype)
    nz = wss > np.finfo(dtype).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2: -(n_fft // 2)]


def stft(y, hp, dtype=np.float32):
    """librosa.stft(y, hp.n_fft, hp.hop_length, win_length=hp.win_length)  (utils.py:101) -> (1+n_fft//2, n_frames)."""
    n_fft, hop = hp.n_fft, hp.hop_length
    w = padded_window(hp, dtype)
    yp = np.pad(y.astype(dtype), n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(yp[idx] * w[None, :], axis=1).T


def denormalize(mag, hp, dtype=np.float32):
    """utils.py:79-86: (T, 1+n_fft//2) network output in [0,1] -> amplitude ** power, transposed to (1+n_fft//2, T)."""
    m = mag.T.astype(dtype)
    m = (np.clip(m, 0, 1) * dtype(hp.max_db)) - dtype(hp.max_db) + dtype(hp.ref_db)
    m = np.power(dtype(10.0), m * dtype(0.05))
    return m ** dtype(hp.power)


def griffin_lim(spec, hp, dtype=np.float32, n_iter=None):
    """utils.py:96-106.  spec: (1+n_fft//2, n_frames) magnitudes."""
    cdtype = np.complex64 if dtype == np.float32 else np.complex128
    n_iter = hp.n_iter if n_iter is None else n_iter
    X_best = spec.astype(cdtype)
    for _ in range(n_iter):
        X_t = istft(X_best, hp, dtype)
        est = stft(X_t, hp, dtype).astype(cdtype)
        phase = est / np.maximum(dtype(1e-8), np.abs(est))
        X_best = (spec * phase).astype(cdtype)
    return np.real(istft(X_best, hp, dtype))


def deemphasis(x, hp, dtype=np.float64):
    """scipy.signal.lfilter([1], [1, -preemphasis], x)  (utils.py:89): y[n] = x[n] + preemphasis * y[n-1]."""
    a = dtype(hp.preemphasis)
    y = np.empty(len(x), dtype)
    prev = dtype(0)
    for n, v in enumerate(x.astype(dtype)):
        prev = v + a * prev
        y[n] = prev
    return y


def frame_rms_power(y, frame_length=2048, hop_length=512, dtype=np.float64):
    """librosa.feature.rmse(y, frame_length, hop_length) ** 2 with center=True, pad_mode='reflect': mean square per frame."""
    yp = np.pad(y.astype(dtype), frame_length // 2, mode="reflect")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    return np.mean(np.abs(yp[idx]) ** 2, axis=1)


def trim_bounds(y, top_db=60, frame_length=2048, hop_length=512, dtype=np.float64):
    """librosa.effects.trim(y)[1] (utils.py:92): [start, end) sample indices of the non-silent region."""
    mse = frame_rms_power(y, frame_length, hop_length, dtype)
    amin = 1e-10
    db = 10.0 * np.log10(np.maximum(amin, mse)) - 10.0 * np.log10(np.maximum(amin, mse.max()))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return 0, 0
    return int(nz[0] * hop_length), min(len(y), int((nz[-1] + 1) * hop_length))


def spectrogram2wav(mag, hp, dtype=np.float32, n_iter=None, return_untrimmed=False):
    """utils.py:67-94.  mag: (T, 1+n_fft//2) in [0,1] -> trimmed float32 waveform."""
    wav = griffin_lim(denormalize(mag, hp, dtype), hp, dtype, n_iter)
    wav = deemphasis(wav, hp, dtype)
    s, e = trim_bounds(wav, dtype=dtype)
    if return_untrimmed:
        return wav.astype(np.float32), (s, e)
    return wav[s:e].astype(np.float32)
