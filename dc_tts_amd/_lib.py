"""ctypes binding of libdctts_hip.so (include/dctts_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``dc_tts_amd/build.py``.  There is NO
fallback: if the shared object is missing or fails to load, importing the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdctts_hip.so")
_DEFAULT_LIB_PATH = LIB_PATH

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t


class Config(ctypes.Structure):
    _fields_ = [("vocab_size", c_int), ("e", c_int), ("d", c_int), ("c", c_int), ("n_mels", c_int),
                ("n_linear", c_int), ("max_N", c_int), ("attention_win_size", c_int)]


class VocoderConfig(ctypes.Structure):
    _fields_ = [("n_fft", c_int), ("hop_length", c_int), ("win_length", c_int), ("n_iter", c_int),
                ("power", ctypes.c_float), ("preemphasis", ctypes.c_float), ("max_db", ctypes.c_float),
                ("ref_db", ctypes.c_float), ("trim_top_db", ctypes.c_float), ("trim_frame_length", c_int),
                ("trim_hop_length", c_int)]


# name -> (restype, argtypes); every symbol include/dctts_hip.h declares
SYMBOLS = {
    "dctts_create": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(Config)]),
    "dctts_destroy": (c_int, [c_void_p]),
    "dctts_last_error": (ctypes.c_char_p, []),
    "dctts_weights_set": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_int64), c_int]),
    "dctts_weights_finalize": (c_int, [c_void_p]),
    "dctts_textenc_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dctts_audioenc_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dctts_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "dctts_audiodec_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dctts_ssrn_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dctts_text2mel_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dctts_synthesize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dctts_decode_status": (c_int, [c_void_p]),
    "dctts_set_team_kernels": (c_int, [c_void_p, c_int]),
    "dctts_decode_safe_once": (c_int, [c_void_p]),
    "dctts_set_workspace_limit": (c_int, [c_void_p, c_size_t]),
    "dctts_set_split_bf16": (c_int, [c_void_p, c_int]),
    "dctts_debug_inject_decode_error": (c_int, [c_void_p, c_int]),
    "dctts_debug_team_kernels_state": (c_int, [c_void_p]),
    "dctts_set_decode_graph": (c_int, [c_void_p, c_int]),
    "dctts_set_decode_mode": (c_int, [c_void_p, c_int]),
    "dctts_device_bytes": (c_size_t, [c_void_p]),
    "dctts_debug_layer": (c_int, [c_void_p, ctypes.c_char_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dctts_debug_seed_prev_max": (c_int, [c_void_p, c_void_p, c_int]),
    "dctts_debug_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dctts_debug_set_trace": (c_int, [c_void_p, c_int, ctypes.c_char_p]),
    "dctts_debug_xcd_census": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "dctts_prof_enable": (c_int, [c_void_p, c_int]),
    "dctts_prof_collect": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]),
    "dctts_prof_rows": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_longlong)]),
    "dctts_vocoder_create": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(VocoderConfig)]),
    "dctts_vocoder_destroy": (c_int, [c_void_p]),
    "dctts_vocoder_device_bytes": (c_size_t, [c_void_p]),
    "dctts_spectrogram2wav": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dctts_griffin_lim": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dctts_vocoder_prof_enable": (c_int, [c_void_p, c_int]),
    "dctts_vocoder_prof_collect": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]),
    # include/dctts_train.h (first training slice)
    "dctts_train_create": (c_int, [ctypes.POINTER(c_void_p), c_int]),
    "dctts_train_destroy": (c_int, [c_void_p]),
    "dctts_train_device_bytes": (c_size_t, [c_void_p]),
    "dctts_train_tape": (c_int, [c_void_p, c_int]),
    "dctts_train_hc_backward": (c_int, [c_void_p] + [c_void_p] * 8 + [c_int] * 6 + [c_void_p] * 7 + [c_void_p]),
    "dctts_train_conv1d_backward": (c_int, [c_void_p] + [c_void_p] * 6 + [c_int] * 8 + [c_void_p] * 5 + [c_void_p]),
    "dctts_train_conv1d_transpose_backward": (c_int, [c_void_p] + [c_void_p] * 6 + [c_int] * 4 + [c_void_p] * 5 + [c_void_p]),
    "dctts_train_attention_backward": (c_int, [c_void_p] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 3 + [c_void_p]),
    "dctts_train_embed_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "dctts_train_hc_forward": (c_int, [c_void_p] + [c_void_p] * 7 + [c_int] * 6 + [c_void_p, c_void_p]),
    "dctts_train_conv1d_forward": (c_int, [c_void_p] + [c_void_p] * 5 + [c_int] * 8 + [c_void_p, c_void_p]),
    "dctts_train_conv1d_transpose_forward": (c_int, [c_void_p] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_void_p]),
    "dctts_train_embed_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "dctts_train_attention_forward": (c_int, [c_void_p] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 2 + [c_void_p]),
    "dctts_train_dropout": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_uint64, ctypes.c_float, c_void_p]),
    "dctts_train_sigmoid": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p]),
    "dctts_train_text2mel_losses": (c_int, [c_void_p] + [c_void_p] * 4 + [c_int] * 6 + [c_void_p] * 4 + [c_void_p]),
    "dctts_train_ssrn_losses": (c_int, [c_void_p] + [c_void_p] * 3 + [ctypes.c_longlong] + [c_void_p] * 3 + [c_void_p]),
    "dctts_train_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, ctypes.c_float, c_void_p]),
    "dctts_train_adam_step_multi": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_float, c_void_p]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError if it is not built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  dc_tts_amd has no CPU or PyTorch fallback path.")
    # torch ships its own libamdhip64; it must be the ONE HIP runtime in the process (loading /opt/rocm's copy
    # first gives this library a second runtime that sees no device), so import torch before dlopen.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if LIB_PATH != _DEFAULT_LIB_PATH and not hasattr(lib, name):
            continue                     # (tools' A/B against a build of an EARLIER commit: newer hooks are simply absent there; the product's own library must have every symbol)
        fn = getattr(lib, name)          # AttributeError here = header / library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().dctts_last_error().decode("utf-8", "replace")
