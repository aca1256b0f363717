"""dc_tts_amd -- MI355X-native DC-TTS synthesis path (Text2Mel autoregressive decode -> SSRN).

Hand-written HIP kernels for gfx950 behind a C ABI (include/dctts_hip.h), with a Python host layer
that mirrors the reference's ``networks.py`` / ``synthesize.py`` surface.  Importing this package does
not need a GPU; constructing an :class:`Engine` does, and fails loudly without the built library.
"""
from .hyperparams import Hyperparams, hp  # noqa: F401

__all__ = ["Hyperparams", "hp"]
