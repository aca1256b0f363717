"""Writes tests/golden/vocoder_seed7.npz: a seeded (F=48, 1025) magnitude map in [0,1] with a speech-like envelope (silence,
a harmonic burst, silence) and oracle/vocoder_ref.spectrogram2wav's float32 result for it (n_iter = 6, untrimmed + bounds).
The reference itself cannot run here (TensorFlow / librosa absent): this pins the RESTATEMENT, not the reference.
Run from the repo root:  python tests/golden/make_golden_vocoder.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dc_tts_amd.hyperparams import hp          # noqa: E402
from oracle import vocoder_ref as V            # noqa: E402


def make_mag(F=48, seed=7):
    rng = np.random.default_rng(seed)
    k = np.arange(hp.n_linear)[None, :]
    f = np.arange(F)[:, None]
    env = np.exp(-((f - F / 2) / (F / 6)) ** 2)                                  # loud in the middle, quiet at both ends
    harm = 0.5 + 0.5 * np.cos(2 * np.pi * k / (40 + 0.2 * f))                    # drifting harmonic comb
    tilt = np.exp(-k / 400.0)
    mag = 0.15 + 0.8 * env * harm * tilt + 0.03 * rng.random((F, hp.n_linear))
    return np.clip(mag, 0, 1).astype(np.float32)


if __name__ == "__main__":
    mag = make_mag()
    n_iter = 6
    wav, (s, e) = V.spectrogram2wav(mag, hp, np.float32, n_iter=n_iter, return_untrimmed=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocoder_seed7.npz")
    np.savez_compressed(out, mag=mag, wav=wav, bounds=np.array([s, e], np.int32), n_iter=np.int32(n_iter))
    print(out, wav.shape, (s, e), float(np.abs(wav).max()), os.path.getsize(out))
