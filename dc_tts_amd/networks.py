"""Drop-in for the reference's ``networks.py`` call surface (networks.py:14-292).

Same function names, argument order and return arity as the reference:

    TextEnc(L, training=True)                      -> (K, V)                      networks.py:14
    AudioEnc(S, training=True)                     -> Q                           networks.py:73
    Attention(Q, K, V, mononotic_attention=False,
              prev_max_attentions=None)            -> (R, alignments, max_attentions)   networks.py:126
    AudioDec(R, training=True)                     -> (logits, Y)                 networks.py:157
    SSRN(Y, training=True)                         -> (logits, Z)                 networks.py:214

(the misspelt ``mononotic_attention`` keyword is the reference's).  Tensors are torch device
tensors, channel-last ``(B, time, C)`` float32; ids / prev_max int32; ``max_attentions`` int64.

The reference finds its weights through TF variable scopes; here the functions use the engine bound
with :func:`bind` (one per process/GPU).  Only inference semantics exist: ``training`` gates dropout
alone in the reference (modules.py:139,195,245), so ``training=True`` raises NotImplementedError.
"""
from typing import Optional

from .engine import Engine

_engine: Optional[Engine] = None


def bind(engine: Engine) -> Engine:
    """Make ``engine`` (weights + GPU) the one the module-level network functions use."""
    global _engine
    _engine = engine
    return engine


def bound_engine() -> Engine:
    if _engine is None:
        raise RuntimeError("dc_tts_amd.networks: no engine bound; call networks.bind(Engine(weights)) first")
    return _engine


def _inference_only(training):
    if training:
        raise NotImplementedError("dc_tts_amd implements the synthesis path only: call with training=False "
                                  "(the reference default training=True enables dropout, which synthesis never uses)")


def TextEnc(L, training=True):
    _inference_only(training)
    return bound_engine().text_enc(L)


def AudioEnc(S, training=True):
    _inference_only(training)
    return bound_engine().audio_enc(S)


def Attention(Q, K, V, mononotic_attention=False, prev_max_attentions=None):
    return bound_engine().attention(Q, K, V, mononotic_attention, prev_max_attentions)


def AudioDec(R, training=True):
    _inference_only(training)
    return bound_engine().audio_dec(R)


def SSRN(Y, training=True):
    _inference_only(training)
    return bound_engine().ssrn(Y)
