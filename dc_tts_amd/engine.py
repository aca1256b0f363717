"""Engine: one libdctts_hip context per GPU, fed with torch device tensors.

PyTorch is plumbing here (device memory for inputs/outputs, the current stream); all arithmetic
runs in the hand-written HIP kernels behind the C ABI (include/dctts_hip.h).
"""
import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .hyperparams import Hyperparams, hp as _hp
from .weights import check_weights


class DcttsError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check(t: torch.Tensor, name: str, dtype, ndim: int, device: torch.device):
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch.Tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype} != {dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    if t.device != device:
        raise ValueError(f"{name}: tensor on {t.device}, engine on {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


class Engine:
    """Owns the device-resident packed weights and workspaces for one GPU."""

    def __init__(self, weights: Dict[str, np.ndarray], hp: Hyperparams = _hp, device: Optional[int] = None,
                 decode_graph: bool = False, keep_weights: bool = True, split_bf16: int = 0):
        if not torch.cuda.is_available():
            raise DcttsError("dc_tts_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self.hp = hp
        check_weights(weights, hp)
        self.weights = weights if keep_weights else None      # host dict (by reference): the training=True forward of networks.py uploads it in TF layout on first use
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        cfg = _lib.Config(len(hp.vocab), hp.e, hp.d, hp.c, hp.n_mels, hp.n_linear, hp.max_N, hp.attention_win_size)
        h = ctypes.c_void_p()
        self._ok(self.lib.dctts_create(ctypes.byref(h), self.device_index, ctypes.byref(cfg)))
        self._h = h
        # split_bf16 (OPT-IN, default 0 = exact fp32): 1 = SSRN, 2 = SSRN + TextEnc contract on the bf16 matrix pipe from split operands (dctts_set_split_bf16);
        # the level given here is what gets packed at upload, set_split_bf16() then selects any level up to it
        if split_bf16:
            self._ok(self.lib.dctts_set_split_bf16(self._h, int(split_bf16)))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._ok(self.lib.dctts_weights_set(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim))
        self._ok(self.lib.dctts_weights_finalize(self._h))
        self.set_decode_graph(decode_graph)

    # ------------------------------------------------------------------ plumbing
    def _ok(self, rc: int):
        if rc != 0:
            raise DcttsError(f"libdctts_hip error {rc}: {_lib.last_error()}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dctts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_split_bf16(self, mode: int):
        """0: exact fp32 (default); 1: SSRN, 2: SSRN + TextEnc on split-bf16 operands -- only levels up to the one the Engine was created with."""
        self._ok(self.lib.dctts_set_split_bf16(self._h, int(mode)))

    def set_decode_graph(self, enable):
        """False/0 (default): eager launches; True/1: the side stream's work as one hipGraph per frame (measured slower)."""
        self._ok(self.lib.dctts_set_decode_graph(self._h, int(bool(enable))))

    def set_decode_mode(self, mode: int):
        """3 (default): two-stream incremental decode (hoisted taps, row-op cone layers); 0: fused full-row kernels on one stream (cross-check)."""
        self._ok(self.lib.dctts_set_decode_mode(self._h, int(mode)))

    def decode_status(self):
        """Raise if a decode on this engine failed on the device (call after synchronising); see dctts_decode_status."""
        self._ok(self.lib.dctts_decode_status(self._h))

    def synchronize(self):
        """Wait for everything issued on the current stream and raise if a decode's bounded in-kernel wait gave up (its outputs are then invalid).
        Call this before trusting / downloading the results of text2mel() or synthesize()."""
        torch.cuda.current_stream(self.device).synchronize()
        self.decode_status()

    def debug_seed_prev_max(self, prev_max):
        """Test hook (dctts_hip_debug.h): the next decode starts from these prev_max_attentions instead of zeros."""
        a = np.ascontiguousarray(np.asarray(prev_max, dtype=np.int32))
        self._ok(self.lib.dctts_debug_seed_prev_max(self._h, a.ctypes.data_as(ctypes.c_void_p), int(a.shape[0])))

    def device_bytes(self) -> int:
        return int(self.lib.dctts_device_bytes(self._h))

    def set_workspace_limit(self, nbytes: int):
        """Workspaces are cached per batch geometry and only grow; past this many bytes the next call drops the cache (one device sync)."""
        self._ok(self.lib.dctts_set_workspace_limit(self._h, int(nbytes)))

    def prof_enable(self, kernel_id: int):
        self._ok(self.lib.dctts_prof_enable(self._h, int(kernel_id)))

    def prof_collect(self) -> Tuple[int, float]:
        n = ctypes.c_int(0); ms = ctypes.c_double(0.0)
        self._ok(self.lib.dctts_prof_collect(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def prof_rows(self) -> int:
        """Output rows covered by the profiled launches since prof_enable (a layer may be split with a 16-row tail launch)."""
        r = ctypes.c_longlong(0)
        self._ok(self.lib.dctts_prof_rows(self._h, ctypes.byref(r)))
        return r.value

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ networks.py surface
    def text_enc(self, L: torch.Tensor):
        _check(L, "L", torch.int32, 2, self.device)
        B, N = L.shape
        K = self._new(B, N, self.hp.d); V = self._new(B, N, self.hp.d)
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return K, V
        self._ok(self.lib.dctts_textenc_fwd(self._h, _ptr(L), B, N, _ptr(K), _ptr(V), self._stream()))
        return K, V

    def audio_enc(self, S: torch.Tensor):
        _check(S, "S", torch.float32, 3, self.device)
        B, T, C = S.shape
        if C != self.hp.n_mels:
            raise ValueError(f"S: last dim {C} != n_mels {self.hp.n_mels}")
        Q = self._new(B, T, self.hp.d)
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return Q
        self._ok(self.lib.dctts_audioenc_fwd(self._h, _ptr(S), B, T, _ptr(Q), self._stream()))
        return Q

    def attention(self, Q, K, V, mononotic_attention=False, prev_max_attentions=None):
        _check(Q, "Q", torch.float32, 3, self.device); _check(K, "K", torch.float32, 3, self.device)
        _check(V, "V", torch.float32, 3, self.device)
        B, T, d = Q.shape
        N = K.shape[1]
        if K.shape != (B, N, d) or V.shape != (B, N, d) or d != self.hp.d:
            raise ValueError(f"Attention: incompatible shapes Q{tuple(Q.shape)} K{tuple(K.shape)} V{tuple(V.shape)}")
        if mononotic_attention:
            if prev_max_attentions is None:
                raise ValueError("Attention: mononotic_attention=True needs prev_max_attentions")
            _check(prev_max_attentions, "prev_max_attentions", torch.int32, 1, self.device)
            if N != self.hp.max_N or T != self.hp.max_T:
                # the reference builds the mask from hp.max_N / hp.max_T (networks.py:142-145): tf.where would fail
                raise ValueError(f"Attention: monotonic mode needs N == hp.max_N ({self.hp.max_N}) and T == hp.max_T "
                                 f"({self.hp.max_T}); got N={N}, T={T}")
        R = self._new(B, T, 2 * d); al = self._new(B, N, T); mx = self._new(B, T, dtype=torch.int64)
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return R, al, mx
        self._ok(self.lib.dctts_attention_fwd(self._h, _ptr(Q), _ptr(K), _ptr(V), B, T, N, int(bool(mononotic_attention)),
                                              _ptr(prev_max_attentions), _ptr(R), _ptr(al), _ptr(mx), self._stream()))
        return R, al, mx

    def audio_dec(self, R: torch.Tensor):
        _check(R, "R", torch.float32, 3, self.device)
        B, T, C = R.shape
        if C != 2 * self.hp.d:
            raise ValueError(f"R: last dim {C} != 2*d")
        logits = self._new(B, T, self.hp.n_mels); Y = self._new(B, T, self.hp.n_mels)
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return logits, Y
        self._ok(self.lib.dctts_audiodec_fwd(self._h, _ptr(R), B, T, _ptr(logits), _ptr(Y), self._stream()))
        return logits, Y

    def ssrn(self, Y: torch.Tensor, want_logits: bool = True):
        _check(Y, "Y", torch.float32, 3, self.device)
        B, T, C = Y.shape
        if C != self.hp.n_mels:
            raise ValueError(f"Y: last dim {C} != n_mels")
        F = self.hp.n_linear
        Z = self._new(B, self.hp.r * T, F)
        logits = self._new(B, self.hp.r * T, F) if want_logits else None
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return logits, Z
        self._ok(self.lib.dctts_ssrn_fwd(self._h, _ptr(Y), B, T, _ptr(logits), _ptr(Z), self._stream()))
        return logits, Z

    def debug_layer(self, net: str, index: int, X: torch.Tensor, cout: int, upsample: int = 1):
        """Test hook: one device layer on X (B,T,Cin) (or int32 ids (B,T) for textenc index 0)."""
        B, T = X.shape[0], X.shape[1]
        out = self._new(B, upsample * T, cout)
        self._ok(self.lib.dctts_debug_layer(self._h, net.encode(), index, _ptr(X.contiguous()), B, T, _ptr(out), self._stream()))
        return out

    # ------------------------------------------------------------------ synthesize.py:45-57
    def _decode(self, call, check: bool):
        """Enqueue a decode; with check=True behave like the reference's `sess.run`: wait for it, and if it failed on the device (a bounded
        in-launch wait gave up: the GPU is shared with somebody else's kernels) repeat it ONCE in the safe form (dctts_decode_safe_once: one
        launch per layer, the streams meet through stream operations -- nothing in it is bounded, so it cannot time out).  The context's
        persistent settings (set_team_kernels, a switch-off by decode_status) are not touched by the retry.  Without check the call returns
        at once; a failed decode's outputs are NaN / -1 and `synchronize()` raises."""
        call()
        if not check:
            return
        torch.cuda.current_stream(self.device).synchronize()
        if self.lib.dctts_decode_status(self._h) == 0:
            return
        first = _lib.last_error()
        self._ok(self.lib.dctts_decode_safe_once(self._h))
        call()
        torch.cuda.current_stream(self.device).synchronize()
        if self.lib.dctts_decode_status(self._h) != 0:
            raise DcttsError(f"decode failed twice: {first} / {_lib.last_error()}")

    def set_team_kernels(self, enable: bool):
        """True (default): runs of dependent decode layers as one launch (xgroup / xcone kernels); False: one launch per layer."""
        self._ok(self.lib.dctts_set_team_kernels(self._h, int(bool(enable))))

    def debug_team_kernels_state(self) -> int:
        """Test hook (dctts_hip_debug.h): bits 1 / 2 = xgroup / xcone requested, 4 = not switched off by a failed status report."""
        return int(self.lib.dctts_debug_team_kernels_state(self._h))

    def debug_inject_decode_error(self, bits: int = 1):
        """Test hook (dctts_hip_debug.h): the next decode reports a failure and poisons its outputs."""
        self._ok(self.lib.dctts_debug_inject_decode_error(self._h, int(bits)))

    def debug_set_trace(self, frame: int, path: Optional[str] = None):
        """Measurement hook (dctts_hip_debug.h): decodes that follow write frame `frame`'s in-kernel stamps to `path`; frame < 0 switches it off."""
        self._ok(self.lib.dctts_debug_set_trace(self._h, int(frame), None if path is None else path.encode()))

    def debug_xcd_census(self):
        """Measurement hook (dctts_hip_debug.h): (XCD of each block of a 128-block launch, compute units of the device)."""
        a = np.zeros(128, np.int32); n = ctypes.c_int(0)
        self._ok(self.lib.dctts_debug_xcd_census(self._h, a.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n), self._stream()))
        return a, n.value

    def text2mel(self, L: torch.Tensor, max_T: Optional[int] = None, alignments: bool = False, check: bool = False):
        """The autoregressive loop of synthesize.py:45-54.  Returns (Y (B,T,n_mels), max_attentions (B,T) int64) and, with
        alignments=True, `g.alignments` (B,N,T) as the loop's last step fetches it (synthesize.py:48)."""
        _check(L, "L", torch.int32, 2, self.device)
        B, N = L.shape
        T = self.hp.max_T if max_T is None else int(max_T)
        if N != self.hp.max_N:
            raise ValueError(f"L must be padded to hp.max_N={self.hp.max_N} (data_load.py:83); got {N}")
        Y = self._new(B, T, self.hp.n_mels); mx = self._new(B, T, dtype=torch.int64)
        al = self._new(B, N, T) if alignments else None
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return (Y, mx, al) if alignments else (Y, mx)
        self._decode(lambda: self._ok(self.lib.dctts_text2mel_decode(self._h, _ptr(L), B, N, T, _ptr(Y), _ptr(mx), _ptr(al), self._stream())), check)
        return (Y, mx, al) if alignments else (Y, mx)

    def synthesize(self, L: torch.Tensor, max_T: Optional[int] = None, alignments: bool = False, check: bool = False):
        """synthesize.py:45-57 without the vocoder: returns (Y, Z (B,4T,1025), max_attentions[, alignments])."""
        _check(L, "L", torch.int32, 2, self.device)
        B, N = L.shape
        T = self.hp.max_T if max_T is None else int(max_T)
        if N != self.hp.max_N:
            raise ValueError(f"L must be padded to hp.max_N={self.hp.max_N} (data_load.py:83); got {N}")
        Y = self._new(B, T, self.hp.n_mels); Z = self._new(B, self.hp.r * T, self.hp.n_linear)
        mx = self._new(B, T, dtype=torch.int64)
        al = self._new(B, N, T) if alignments else None
        if B == 0:                                                           # an empty batch: what sess.run returns for an empty feed (arrays with a zero leading dimension); nothing to launch
            return (Y, Z, mx, al) if alignments else (Y, Z, mx)
        self._decode(lambda: self._ok(self.lib.dctts_synthesize(self._h, _ptr(L), B, N, T, _ptr(Y), _ptr(Z), _ptr(mx), _ptr(al), self._stream())), check)
        return (Y, Z, mx, al) if alignments else (Y, Z, mx)
