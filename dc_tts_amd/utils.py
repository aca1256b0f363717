"""The vocoder tail of the reference's `utils.py` on the GPU: `spectrogram2wav` (utils.py:67-94), `griffin_lim`
(utils.py:96-106) and `invert_spectrogram` (utils.py:108-114), batched.

Same names and argument meaning as the reference, with two differences that the batching forces:
  * spectrograms are (T, 1 + n_fft//2) per utterance exactly as `synthesize.py:61-63` hands them over (`mag`), or a
    batch (B, T, 1 + n_fft//2); the reference's internal transpose to librosa's (1 + n_fft//2, T) never happens;
  * batched calls return a list of trimmed float32 waveforms (utterances trim to different lengths).
All arithmetic runs in libdctts_hip.so (csrc/vocoder_kernels.h); there is no CPU fallback.
"""
import ctypes
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .engine import DcttsError, _check, _ptr
from .hyperparams import Hyperparams, hp as _hp

TRIM_TOP_DB, TRIM_FRAME_LENGTH, TRIM_HOP_LENGTH = 60.0, 2048, 512     # librosa.effects.trim defaults (utils.py:92)


class Vocoder:
    """One vocoder handle per GPU (no weights; workspaces grow to the largest batch seen)."""

    def __init__(self, hp: Hyperparams = _hp, device: Optional[int] = None):
        if not torch.cuda.is_available():
            raise DcttsError("dc_tts_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self.hp = hp
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        cfg = _lib.VocoderConfig(hp.n_fft, hp.hop_length, hp.win_length, hp.n_iter, hp.power, hp.preemphasis,
                                 float(hp.max_db), float(hp.ref_db), TRIM_TOP_DB, TRIM_FRAME_LENGTH, TRIM_HOP_LENGTH)
        h = ctypes.c_void_p()
        self._ok(self.lib.dctts_vocoder_create(ctypes.byref(h), self.device_index, ctypes.byref(cfg)))
        self._h = h

    def _ok(self, rc: int):
        if rc != 0:
            raise DcttsError(f"libdctts_hip error {rc}: {_lib.last_error()}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dctts_vocoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return int(self.lib.dctts_vocoder_device_bytes(self._h))

    def prof_enable(self, enable: bool):
        self._ok(self.lib.dctts_vocoder_prof_enable(self._h, int(bool(enable))))

    def prof_collect(self) -> Tuple[int, float]:
        """(launches, total ms) of the Griffin-Lim iteration kernel since prof_enable(True), by HIP events on the launch stream."""
        n = ctypes.c_int(0); ms = ctypes.c_double(0.0)
        self._ok(self.lib.dctts_vocoder_prof_collect(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def n_samples(self, F: int) -> int:
        """Length of librosa.istft's output for F frames: hop_length * (F - 1)."""
        return self.hp.hop_length * (F - 1)

    def spectrogram2wav_device(self, mag: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """mag (B,F,1+n_fft//2) on the GPU -> (wav (B, hop*(F-1)) de-emphasised and untrimmed, bounds (B,2) int32)."""
        _check(mag, "mag", torch.float32, 3, self.device)
        B, F, nb = mag.shape
        if nb != self.hp.n_linear:
            raise ValueError(f"mag: last dim {nb} != 1 + n_fft//2 = {self.hp.n_linear}")
        wav = torch.empty(B, self.n_samples(F), dtype=torch.float32, device=self.device)
        bounds = torch.empty(B, 2, dtype=torch.int32, device=self.device)
        self._ok(self.lib.dctts_spectrogram2wav(self._h, _ptr(mag), B, F, _ptr(wav), _ptr(bounds), self._stream()))
        return wav, bounds

    def griffin_lim_device(self, spec: torch.Tensor, n_iter: Optional[int] = None, want_X: bool = False):
        """spec (B,F,1+n_fft//2) magnitudes on the GPU -> y (B, hop*(F-1)) [, X_best (B,F,1+n_fft//2) complex64]."""
        _check(spec, "spec", torch.float32, 3, self.device)
        B, F, nb = spec.shape
        if nb != self.hp.n_linear:
            raise ValueError(f"spec: last dim {nb} != 1 + n_fft//2 = {self.hp.n_linear}")
        n_iter = self.hp.n_iter if n_iter is None else int(n_iter)
        y = torch.empty(B, self.n_samples(F), dtype=torch.float32, device=self.device)
        X = torch.empty(B, F, nb, 2, dtype=torch.float32, device=self.device) if want_X else None
        self._ok(self.lib.dctts_griffin_lim(self._h, _ptr(spec), B, F, n_iter, _ptr(y), _ptr(X), self._stream()))
        return (y, torch.view_as_complex(X)) if want_X else y


_default: dict = {}


def _vocoder(hp: Hyperparams, device: Optional[int]) -> Vocoder:
    key = (hp, torch.cuda.current_device() if device is None and torch.cuda.is_available() else device)
    if key not in _default:
        _default[key] = Vocoder(hp, device)
    return _default[key]


def _to_dev(a, device: torch.device) -> Tuple[torch.Tensor, bool]:
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) if isinstance(a, np.ndarray) else a
    single = t.dim() == 2
    if single:
        t = t[None]
    return t.to(device=device, dtype=torch.float32).contiguous(), single


def spectrogram2wav(mag, hp: Hyperparams = _hp, vocoder: Optional[Vocoder] = None):
    """utils.py:67-94.  mag: (T, 1+n_fft//2) -> 1-D float32 numpy waveform (trimmed), as the reference returns it;
    or a batch (B, T, 1+n_fft//2) (numpy or torch) -> list of B such waveforms."""
    v = vocoder or _vocoder(hp, None)
    m, single = _to_dev(mag, v.device)
    wav, bounds = v.spectrogram2wav_device(m)
    wav_h, b_h = wav.cpu().numpy(), bounds.cpu().numpy()
    out: List[np.ndarray] = [wav_h[i, b_h[i, 0]:b_h[i, 1]].copy() for i in range(wav_h.shape[0])]
    return out[0] if single else out


def griffin_lim(spectrogram, hp: Hyperparams = _hp, vocoder: Optional[Vocoder] = None):
    """utils.py:96-106, on (T, 1+n_fft//2) magnitudes (or a batch): returns the real waveform(s) as numpy."""
    v = vocoder or _vocoder(hp, None)
    s, single = _to_dev(spectrogram, v.device)
    y = v.griffin_lim_device(s).cpu().numpy()
    return y[0] if single else y


def invert_spectrogram(spectrogram, hp: Hyperparams = _hp, vocoder: Optional[Vocoder] = None):
    """utils.py:108-114 (librosa.istft with the Hann window) for a REAL (T, 1+n_fft//2) spectrogram (zero phase) --
    the only real-input use the reference has is griffin_lim's first pass."""
    v = vocoder or _vocoder(hp, None)
    s, single = _to_dev(spectrogram, v.device)
    y = v.griffin_lim_device(s, n_iter=0).cpu().numpy()
    return y[0] if single else y
