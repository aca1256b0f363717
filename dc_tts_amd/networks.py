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
with :func:`bind` (one per process/GPU).

``training`` gates dropout alone in the reference (modules.py:139,195,245).  ``training=False`` (what
synthesize.py builds its graph with, train.py:33) runs the fused inference kernels.  ``training=True``
-- the reference's DEFAULT -- runs the training forward pass of include/dctts_train.h on the same
weights: every block followed by dropout(hp.dropout_rate) drawn from a counter-based hash of
(seed, call number, layer, element).  TensorFlow's random stream cannot be reproduced; the mask is the
one oracle/train_ref.py restates, so a training=True call is checkable against the oracle
(tests/test_gpu_train.py), and its expectation equals the training=False result.
"""
from typing import Optional

import numpy as np
import torch

from .engine import Engine

_engine: Optional[Engine] = None
_training = None            # (engine, TrainOps, {TF variable name: device tensor}): built on the first training=True call
_training_seed = 0
_training_calls = 0


def bind(engine: Engine) -> Engine:
    """Make ``engine`` (weights + GPU) the one the module-level network functions use."""
    global _engine, _training
    _engine = engine
    _training = None
    return engine


def set_training_seed(seed: int) -> None:
    """Seed of the dropout masks of training=True calls (the call counter restarts)."""
    global _training_seed, _training_calls
    _training_seed, _training_calls = int(seed), 0


def _training_forward(layers, prefix: str, x, padding: str):
    """One network with dropout behind every block (modules.py:139,195,245) on the bound engine's weights."""
    global _training, _training_calls
    from .train import TrainOps, network_forward
    eng = bound_engine()
    if eng.weights is None:
        raise RuntimeError("training=True needs the engine's host copy of the weights (Engine(..., keep_weights=True), the default)")
    if _training is None or _training[0] is not eng:
        ops = TrainOps(eng.device_index)
        W = {n: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(eng.device) for n, v in eng.weights.items()}
        _training = (eng, ops, W)
    _, ops, W = _training
    drop = (eng.hp.dropout_rate, _training_seed, _training_calls) if eng.hp.dropout_rate > 0 else None
    _training_calls += 1
    y, _ = network_forward(ops, layers, W, prefix, x, padding, drop)
    return y, ops


def bound_engine() -> Engine:
    if _engine is None:
        raise RuntimeError("dc_tts_amd.networks: no engine bound; call networks.bind(Engine(weights)) first")
    return _engine


def TextEnc(L, training=True):
    if training:
        from .layers import textenc_layers
        hp = bound_engine().hp
        KV, _ = _training_forward(textenc_layers(hp), "Text2Mel/TextEnc", L, "SAME")
        return KV[..., :hp.d].contiguous(), KV[..., hp.d:].contiguous()                 # networks.py:70
    return bound_engine().text_enc(L)


def AudioEnc(S, training=True):
    if training:
        from .layers import audioenc_layers
        return _training_forward(audioenc_layers(bound_engine().hp), "Text2Mel/AudioEnc", S, "CAUSAL")[0]
    return bound_engine().audio_enc(S)


def Attention(Q, K, V, mononotic_attention=False, prev_max_attentions=None):
    return bound_engine().attention(Q, K, V, mononotic_attention, prev_max_attentions)


def AudioDec(R, training=True):
    if training:
        from .layers import audiodec_layers
        logits, ops = _training_forward(audiodec_layers(bound_engine().hp), "Text2Mel/AudioDec", R, "CAUSAL")
        return logits, ops.sigmoid(logits)                                               # networks.py:210
    return bound_engine().audio_dec(R)


def SSRN(Y, training=True):
    if training:
        from .layers import ssrn_layers
        logits, ops = _training_forward(ssrn_layers(bound_engine().hp), "SSRN", Y, "SAME")
        return logits, ops.sigmoid(logits)                                               # networks.py:291
    return bound_engine().ssrn(Y)
