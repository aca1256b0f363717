"""Training input side (CPU): dc_tts_amd.audio (utils.py:18-65,147-165), data_load.load_data("train") / get_batch
(data_load.py:33-140) and prepo (prepo.py).  librosa is not installed: the product's vectorised numpy is checked against the oracle's
independent frame-by-frame restatements, closed forms, and the vocoder round trip -- parity with librosa itself is unpinned."""
import os

import numpy as np
import pytest

from dc_tts_amd import audio as A
from dc_tts_amd import data_load as D
from dc_tts_amd.hyperparams import hp
from oracle import vocoder_ref as VR


def _voice(seconds=1.3, sr=22050, seed=0, lead=0.25, tail=0.2):
    """A harmonic tone with vibrato between two stretches of near-silence (something trim has to find)."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr); t = np.arange(n) / sr
    f0 = 140.0 + 30.0 * np.sin(2 * np.pi * 3.0 * t)
    ph = 2 * np.pi * np.cumsum(f0) / sr
    y = sum(np.sin(k * ph) / k for k in range(1, 9)) * 0.2
    y *= np.minimum(1.0, np.minimum(t / 0.02, (seconds - t) / 0.02))
    pad = lambda s: rng.normal(0, 1e-5, int(s * sr))
    return np.concatenate([pad(lead), y, pad(tail)]).astype(np.float32)


def _write_wav(path, y, sr=22050):
    from scipy.io import wavfile
    wavfile.write(path, sr, np.round(np.clip(y, -1, 1 - 1 / 32768) * 32768).astype(np.int16))


def test_mel_filterbank_slaney_properties():
    M = A.mel_filterbank(22050, 2048, 80)
    assert M.shape == (80, 1025) and M.dtype == np.float32 and (M >= 0).all()
    # librosa's documented example, librosa.filters.mel(sr=22050, n_fft=2048) (128 bands): second bin of the first band prints as 0.016
    assert abs(float(A.mel_filterbank(22050, 2048, 128)[0, 1]) - 0.016) < 5e-4
    # below 1 kHz the Slaney scale is linear (66.67 Hz per mel): equal-width triangles of equal height; above, widths grow and heights fall
    hz = np.linspace(0, 11025, 1025)
    centre = (M * hz[None, :]).sum(1) / M.sum(1)
    assert np.all(np.diff(centre) > 0)
    peak = M.max(1)
    low = centre < 900
    assert low.sum() > 10 and np.allclose(peak[low], peak[low].mean(), rtol=0.08) and np.all(np.diff(peak[-25:]) < 0) and peak[-1] < peak[low].mean() / 8
    # area normalisation: every triangle integrates to 1 over frequency (2 / width * width / 2), up to the 10.77 Hz bin raster
    area = M.sum(1) * (11025 / 1024)
    assert np.allclose(area, 1.0, atol=0.08)
    # every band overlaps its neighbours only: bin support is contiguous and neighbours' supports meet
    for i in range(80):
        nz = np.flatnonzero(M[i]); assert nz.size and np.all(np.diff(nz) == 1)


def test_stft_and_trim_match_the_frame_by_frame_restatement():
    y = _voice()
    S = A.stft(y, hp)
    Sr = VR.stft(y, hp, dtype=np.float64)
    assert S.shape == (1025, 1 + len(y) // hp.hop_length) == Sr.shape
    assert np.abs(S - Sr).max() < 2e-4 * np.abs(Sr).max()
    yt, (s, e) = A.trim(y)
    assert (s, e) == tuple(VR.trim_bounds(y)) and len(yt) == e - s
    lead, tail = int(0.25 * 22050), int(0.2 * 22050)
    assert 0 < s <= lead and len(y) - tail <= e < len(y)                  # cuts into the silence, never into the tone
    assert abs(s - lead) <= 2048 and abs(e - (len(y) - tail)) <= 2048
    assert A.trim(np.zeros(5000, np.float32))[1] == (0, 5000) or A.trim(np.zeros(5000, np.float32))[1] == (0, 0)


def test_stationary_sine_closed_form():
    """A sine of amplitude a at a bin centre: |STFT| = a sum(window) / 2 at that bin; pre-emphasis scales it by |1 - 0.97 e^{-jw}|;
    the normalised dB value follows utils.py:52-58."""
    k = 100
    f = k * hp.sr / hp.n_fft
    n = np.arange(3 * hp.sr // 2)
    a = 0.01                                                                 # small enough that the line stays below the clip at 1
    y = (a * np.sin(2 * np.pi * f * n / hp.sr)).astype(np.float32)
    mel, mag = A.spectrograms_of(y, hp)
    assert mel.shape[1] == 80 and mag.shape[1] == 1025 and mel.shape[0] == mag.shape[0]
    w = 2 * np.pi * f / hp.sr
    amp = a * A.padded_window(hp).sum() / 2 * abs(1 - hp.preemphasis * np.exp(-1j * w))
    want = np.clip((20 * np.log10(amp) - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1)
    mid = mag[5:-5]
    assert np.abs(mid[:, k] - want).max() < 2e-3
    assert np.all(mid.argmax(1) == k)
    assert mid[:, 400:].max() < want - 0.5                                   # > 50 dB down far from the line
    assert mel.min() >= 1e-8 and mel.max() <= 1 and mag.min() >= 1e-8 and mag.max() <= 1 and mel.dtype == np.float32


def test_load_spectrograms_reduction_and_vocoder_round_trip(tmp_path):
    y = _voice(seed=3)
    f = str(tmp_path / "LJ001-0001.wav")
    _write_wav(f, y)
    assert np.abs(A.load_wav(f, 22050) - y).max() < 1.0 / 32768
    mel, mag = A.get_spectrograms(f, hp)
    fname, rmel, pmag = A.load_spectrograms(f, hp)
    assert fname == "LJ001-0001.wav"
    T = mag.shape[0]
    Tp = (T + hp.r - 1) // hp.r * hp.r
    assert pmag.shape == (Tp, 1025) and rmel.shape == (Tp // hp.r, 80)
    np.testing.assert_array_equal(pmag[:T], mag); assert not pmag[T:].any()
    np.testing.assert_array_equal(rmel, np.pad(mel, [[0, Tp - T], [0, 0]])[::hp.r])
    # resampling path: the same tone stored at 44.1 kHz gives nearly the same spectrogram
    f2 = str(tmp_path / "hi.wav")
    from scipy.signal import resample_poly
    _write_wav(f2, resample_poly(y, 2, 1), 44100)
    mel2, _ = A.get_spectrograms(f2, hp)
    n = min(len(mel), len(mel2))
    assert abs(len(mel) - len(mel2)) <= 2 and np.abs(mel[:n] - mel2[:n]).mean() < 0.02
    # the oracle's spectrogram2wav (utils.py:67-96, power 1 instead of hp.power so that magnitudes are comparable) inverts `mag`:
    # the spectrogram of the reconstruction is close to the spectrogram it was made from, in the bins that carry the signal
    h1 = hp.replace(power=1.0, n_iter=30)
    wav = VR.spectrogram2wav(mag, h1, dtype=np.float64)
    _, mag_rt = A.spectrograms_of(wav.astype(np.float32), hp)
    n = min(len(mag), len(mag_rt))
    loud = mag[5:n - 5] > 0.6
    assert loud.sum() > 1000 and np.abs(mag_rt[5:n - 5][loud] - mag[5:n - 5][loud]).mean() < 0.05


def _corpus(tmp_path, n=23, lj=True):
    rng = np.random.default_rng(5)
    root = tmp_path / ("LJSpeech-1.0" if lj else "kate")
    (root / "wavs").mkdir(parents=True)
    words = "the quick brown fox jumps over a lazy dog while seven wizards quietly box".split()
    rows = []
    for i in range(n):
        text = " ".join(rng.choice(words, int(rng.integers(2, 30)))).capitalize() + "."
        name = f"LJ{i:03d}"
        _write_wav(str(root / "wavs" / (name + ".wav")), _voice(seconds=0.25 + 0.04 * (i % 5), seed=i, lead=0.05, tail=0.05))
        rows.append(f"{name}|{text}|{text}" if lj else f"wavs/{name}.wav|x|{text.lower()}|0|{3.0 if i % 7 else 12.5}")
    (root / "transcript.csv").write_text("\n".join(rows) + "\n", encoding="utf-8")
    return str(root), rows


def test_load_data_train_both_corpus_layouts(tmp_path):
    root, rows = _corpus(tmp_path)
    h = hp.replace(data=root)
    fpaths, lens, texts = D.load_data("train", hp=h)
    _, i2c = D.load_vocab()
    assert len(fpaths) == len(rows) and fpaths[3] == os.path.join(root, "wavs", "LJ003.wav")
    for row, n, t in zip(rows, lens, texts):
        want = D.text_normalize(row.split("|")[2]) + "E"
        assert t.dtype == np.int32 and n == len(t) == len(want) and "".join(i2c[int(c)] for c in t) == want
    root2, rows2 = _corpus(tmp_path, lj=False)
    fp2, lens2, texts2 = D.load_data("train", hp=hp.replace(data=root2))
    keep = [r for r in rows2 if float(r.split("|")[4]) <= 10.0]
    assert len(fp2) == len(keep) < len(rows2) and fp2[0] == os.path.join(root2, keep[0].split("|")[0])
    assert "".join(i2c[int(c)] for c in texts2[0]) == keep[0].split("|")[2] + "E"          # no normalisation on this branch


def test_prepo_and_bucketed_batches(tmp_path):
    from dc_tts_amd import prepo
    root, rows = _corpus(tmp_path)
    out = str(tmp_path / "pre")
    h = hp.replace(data=root, B=4)
    assert prepo.main(["--out", out], hp=h) == 0
    assert sorted(os.listdir(os.path.join(out, "mels"))) == sorted(os.listdir(os.path.join(out, "mags"))) == [f"LJ{i:03d}.npy" for i in range(len(rows))]
    out2 = str(tmp_path / "pre2")                                           # the same over worker processes
    assert prepo.main(["--out", out2, "--workers", "2", "--data", root], hp=hp.replace(B=4)) == 0
    for sub in ("mels", "mags"):
        for f in os.listdir(os.path.join(out, sub)):
            np.testing.assert_array_equal(np.load(os.path.join(out, sub, f)), np.load(os.path.join(out2, sub, f)))
    q = D.get_batch(h, seed=1, prepro_dir=out, pad_text_to=4)
    assert q.num_batch == len(rows) // 4
    lens = np.array(q.text_lengths)
    assert q.boundaries == list(range(lens.min() + 1, lens.max() - 1, 20))
    for n in (lens.min(), lens.max(), q.boundaries[0] - 1, q.boundaries[0]):
        b = q.which_bucket(int(n))
        lo = -10 ** 9 if b == 0 else q.boundaries[b - 1]
        hi = 10 ** 9 if b == len(q.boundaries) else q.boundaries[b]
        assert lo <= n < hi
    seen, it = [], iter(q)
    for _ in range(12):                                                     # more than two epochs of 5 batches
        texts, mels, mags, fnames = next(it)
        assert texts.shape[0] == mels.shape[0] == mags.shape[0] == len(fnames) == 4 and texts.dtype == np.int32
        assert texts.shape[1] % 4 == 0 and mags.shape[1] == hp.r * mels.shape[1] and mels.shape[2] == 80 and mags.shape[2] == 1025
        bks = set()
        for i, f in enumerate(fnames):
            k = int(f[2:5])
            n = q.text_lengths[k]
            np.testing.assert_array_equal(texts[i, :n], q.texts[k]); assert not texts[i, n:].any() and texts[i, n - 1] == 1
            m = np.load(os.path.join(out, "mels", f.replace("wav", "npy")))
            np.testing.assert_array_equal(mels[i, :len(m)], m); assert not mels[i, len(m):].any()
            bks.add(q.which_bucket(n))
        assert len(bks) == 1                                                # a batch comes from ONE length bucket
        assert texts.shape[1] - max(q.text_lengths[int(f[2:5])] for f in fnames) < 4
        assert mels.shape[1] == max(len(np.load(os.path.join(out, "mels", f.replace("wav", "npy")))) for f in fnames)
        seen += fnames
    assert len(set(seen)) > len(rows) // 2
    again = [next(iter(D.get_batch(h, seed=1, prepro_dir=out)))[3] for _ in range(1)]
    assert again[0] == seen[:4]                                             # same seed, same stream
    # hp.prepro = False reads the wave files directly and gives the same arrays
    t2, m2, g2, f2 = next(iter(D.get_batch(h.replace(prepro=False), seed=1)))
    assert f2 == seen[:4]
    m1 = np.load(os.path.join(out, "mels", f2[0].replace("wav", "npy")))
    np.testing.assert_array_equal(m2[0, :len(m1)], m1)


def test_against_transformers_audio_utils():
    """A third-party pin: `transformers.audio_utils` (installed here) implements librosa-compatible mel filters (norm="slaney",
    mel_scale="slaney"), windows and spectrograms independently of this repo.  Filterbank, padded window and |STFT| must agree with it."""
    AU = pytest.importorskip("transformers.audio_utils")
    M = A.mel_filterbank(hp.sr, hp.n_fft, hp.n_mels)
    R = AU.mel_filter_bank(1 + hp.n_fft // 2, hp.n_mels, 0.0, hp.sr / 2.0, hp.sr, norm="slaney", mel_scale="slaney").T
    assert np.abs(M - R).max() < 1e-7
    w = AU.window_function(hp.win_length, "hann", periodic=True, frame_length=hp.n_fft, center=True)
    assert np.abs(w - A.padded_window(hp)).max() < 1e-12
    y = _voice(seed=8)
    S = AU.spectrogram(y, w, frame_length=hp.n_fft, hop_length=hp.hop_length, fft_length=hp.n_fft, power=1.0, center=True, pad_mode="reflect")
    mine = np.abs(A.stft(y, hp))
    assert S.shape == mine.shape and np.abs(S - mine).max() < 1e-5 * mine.max()
    # the whole chain of utils.py:41-58 rebuilt from that library's pieces
    yt, _ = A.trim(y)
    ye = np.append(yt[0], yt[1:] - np.float32(hp.preemphasis) * yt[:-1])
    mag = AU.spectrogram(ye, w, frame_length=hp.n_fft, hop_length=hp.hop_length, fft_length=hp.n_fft, power=1.0, center=True, pad_mode="reflect")
    mel = R.astype(np.float32) @ mag
    norm = lambda x: np.clip((20 * np.log10(np.maximum(1e-5, x)) - hp.ref_db + hp.max_db) / hp.max_db, 1e-8, 1).T
    mel_p, mag_p = A.spectrograms_of(y, hp)
    assert np.abs(norm(mag) - mag_p).max() < 1e-4 and np.abs(norm(mel) - mel_p).max() < 1e-4
