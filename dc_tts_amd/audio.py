"""Waveform -> the training targets of train.py: `get_spectrograms` (utils.py:18-65) and `load_spectrograms` (utils.py:147-165).

Host-side numpy, run once per corpus by `dc_tts_amd.prepo` (prepo.py) exactly as the reference does; the inverse direction
(`spectrogram2wav`) is the GPU vocoder in `dc_tts_amd.utils`.  The reference delegates to librosa, which is not installed here and not
vendored in /root/reference (no version pin; 2018, so librosa 0.5 / 0.6): the functions below restate the published algorithms
with that era's defaults and say which.  PARITY UNPINNED against librosa itself: tests check them against independent restatements
(the test suite's frame-by-frame STFT and trim), closed forms (a stationary sine), the vocoder round trip, and a third-party
librosa-compatible implementation that is installed here (transformers.audio_utils: mel filters to 1e-9, |STFT| to 1e-7).

  librosa.load(fpath, sr=hp.sr)          -> load_wav: PCM / float WAV via scipy.io.wavfile, mono (channel mean), float32 in [-1, 1);
                                            a file at another rate is resampled with a polyphase filter (librosa: resampy kaiser_best -- not
                                            bit-compatible; LJ Speech is 22 050 Hz already, so the path is not taken on the reference's corpus)
  librosa.effects.trim(y)                -> trim: top_db = 60 below the loudest frame, frame_length 2048, hop_length 512, centred frames
  librosa.stft(y, n_fft, hop, win)       -> stft: periodic Hann of win_length zero-padded (centred) to n_fft, center=True with reflect padding
  librosa.filters.mel(sr, n_fft, n_mels) -> mel_filterbank: Slaney scale (htk=False), fmin 0, fmax sr/2, area-normalised triangles (norm=1)
"""
import os
from typing import Tuple

import numpy as np

from .hyperparams import Hyperparams, hp as _hp


def load_wav(fpath: str, sr: int) -> np.ndarray:
    from scipy.io import wavfile
    file_sr, y = wavfile.read(fpath)
    if y.dtype == np.int16: y = y.astype(np.float32) / 32768.0
    elif y.dtype == np.int32: y = y.astype(np.float32) / 2147483648.0
    elif y.dtype == np.uint8: y = (y.astype(np.float32) - 128.0) / 128.0
    else: y = y.astype(np.float32)
    if y.ndim == 2: y = y.mean(axis=1, dtype=np.float32)
    if file_sr != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(file_sr), int(sr))
        y = resample_poly(y, sr // g, file_sr // g).astype(np.float32)
    return np.ascontiguousarray(y, dtype=np.float32)


def _frames(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """(n_frames, frame_length) strided view of y: frame t starts at sample t * hop_length."""
    n = 1 + (len(y) - frame_length) // hop_length
    return np.lib.stride_tricks.as_strided(y, shape=(n, frame_length), strides=(y.strides[0] * hop_length, y.strides[0]), writeable=False)


def trim(y: np.ndarray, top_db: float = 60.0, frame_length: int = 2048, hop_length: int = 512) -> Tuple[np.ndarray, Tuple[int, int]]:
    """librosa.effects.trim: frames whose RMS power is within top_db of the loudest frame are signal; keep first .. last of them."""
    yp = np.pad(y.astype(np.float64), frame_length // 2, mode="reflect") if len(y) > 1 else np.zeros(frame_length + 1)
    power = np.mean(_frames(yp, frame_length, hop_length) ** 2, axis=1)          # rms ** 2 per centred frame
    ref = power.max()
    db = 10.0 * np.log10(np.maximum(1e-10, power)) - 10.0 * np.log10(np.maximum(1e-10, ref))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[:0], (0, 0)
    start, end = int(nz[0]) * hop_length, min(len(y), (int(nz[-1]) + 1) * hop_length)
    return y[start:end], (start, end)


def padded_window(hp: Hyperparams) -> np.ndarray:
    n = np.arange(hp.win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / hp.win_length)                       # scipy.signal.get_window("hann", win, fftbins=True)
    lpad = (hp.n_fft - hp.win_length) // 2
    return np.pad(w, (lpad, hp.n_fft - hp.win_length - lpad))


def stft(y: np.ndarray, hp: Hyperparams = _hp) -> np.ndarray:
    """(1 + n_fft // 2, 1 + len(y) // hop) complex64, frame t centred on sample t * hop."""
    yp = np.pad(y.astype(np.float64), hp.n_fft // 2, mode="reflect")
    fr = _frames(yp, hp.n_fft, hp.hop_length) * padded_window(hp)[None, :]
    return np.fft.rfft(fr, axis=1).T.astype(np.complex64)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)      # above 1 kHz the Slaney scale is logarithmic
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3.0) * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """(n_mels, 1 + n_fft // 2) float32 triangles on the Slaney mel scale, each scaled by 2 / (its band's width in Hz)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def spectrograms_of(y: np.ndarray, hp: Hyperparams = _hp) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:35-65 on samples already loaded: trim, pre-emphasis, |STFT|, mel projection, dB, normalisation to [1e-8, 1]."""
    y, _ = trim(y)
    if len(y) == 0:
        raise ValueError("get_spectrograms: nothing left after trimming (silent file)")
    y = np.append(y[0], y[1:] - np.float32(hp.preemphasis) * y[:-1])               # utils.py:38
    mag = np.abs(stft(y, hp))                                                      # (1 + n_fft // 2, T)
    mel = np.dot(mel_filterbank(hp.sr, hp.n_fft, hp.n_mels), mag)                  # (n_mels, T)
    mel = 20 * np.log10(np.maximum(1e-5, mel))
    mag = 20 * np.log10(np.maximum(1e-5, mag))
    mel = np.clip((mel - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1)
    mag = np.clip((mag - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1)
    return mel.T.astype(np.float32), mag.T.astype(np.float32)


def get_spectrograms(fpath: str, hp: Hyperparams = _hp) -> Tuple[np.ndarray, np.ndarray]:
    """mel (T, n_mels), mag (T, 1 + n_fft // 2), float32 (utils.py:18-65)."""
    return spectrograms_of(load_wav(fpath, hp.sr), hp)


def reduce_frames(mel: np.ndarray, mag: np.ndarray, hp: Hyperparams = _hp) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:155-164: zero frames up to a multiple of r, then every r-th mel frame."""
    t = mel.shape[0]
    num_paddings = hp.r - (t % hp.r) if t % hp.r != 0 else 0
    mel = np.pad(mel, [[0, num_paddings], [0, 0]], mode="constant")
    mag = np.pad(mag, [[0, num_paddings], [0, 0]], mode="constant")
    return mel[::hp.r, :], mag


def load_spectrograms(fpath: str, hp: Hyperparams = _hp):
    """fname, mel (T / r, n_mels), mag (T, 1 + n_fft // 2) with T padded to a multiple of r (utils.py:147-165)."""
    mel, mag = reduce_frames(*get_spectrograms(fpath, hp), hp)
    return os.path.basename(fpath), mel, mag
