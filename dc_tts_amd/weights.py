"""Weights container keyed by the reference's TF variable names.

The reference creates its variables implicitly (``tf.get_variable`` /
``tf.layers``) and restores them by name (``synthesize.py:32-40``).  Here the
weights are an explicit ``dict[str, np.ndarray(float32)]`` with exactly those
names and shapes (``layers.variable_shapes``), so a TF tensor-bundle reader can
drop in later.  No network is available for the pretrained checkpoint
(``README.md:57``), so ``synthetic_weights`` draws seeded weights from the
reference's own initialisers:

  conv / deconv kernels  variance_scaling_initializer() defaults
                         = truncated normal, stddev sqrt(1.3 * 2 / fan_in)
                         (modules.py:132,185,238)
  embedding              truncated normal stddev 0.1 (modules.py:35)
  conv bias 0, LN gamma 1, beta 0 (tf.layers / contrib defaults)

plus, with ``perturb=True`` (the default for parity work), small random
bias / beta and gamma = 1 + N(0, 0.1) so every term of every kernel epilogue is
exercised.
"""
from typing import Dict

import numpy as np

from .hyperparams import Hyperparams, hp as _hp
from .layers import variable_shapes


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():  # TF truncated_normal: redraw anything beyond 2 sigma
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def synthetic_weights(hp: Hyperparams = _hp, seed: int = 1234, perturb: bool = True) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    W: Dict[str, np.ndarray] = {}
    for name, shape in variable_shapes(hp).items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "lookup_table":
            W[name] = _trunc_normal(rng, shape, 0.1)
        elif leaf == "kernel":
            if "conv2d_transpose" in name:          # (1, k, Cout, Cin): TF fan_in = shape[-2] * receptive
                fan_in = shape[1] * shape[2]
            else:                                   # (k, Cin, Cout)
                fan_in = shape[0] * shape[1]
            W[name] = _trunc_normal(rng, shape, np.sqrt(1.3 * 2.0 / fan_in))
        elif leaf == "gamma":
            g = np.ones(shape, np.float32)
            if perturb:
                g = g + 0.1 * rng.standard_normal(shape).astype(np.float32)
            W[name] = g.astype(np.float32)
        elif leaf in ("bias", "beta"):
            b = np.zeros(shape, np.float32)
            if perturb:
                b = 0.1 * rng.standard_normal(shape).astype(np.float32)
            W[name] = b.astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(name)
    return W


def check_weights(W: Dict[str, np.ndarray], hp: Hyperparams = _hp) -> None:
    """Raise ValueError unless W holds every synthesis-path variable with the right shape/dtype."""
    spec = variable_shapes(hp)
    missing = [n for n in spec if n not in W]
    if missing:
        raise ValueError(f"weights missing {len(missing)} variables, e.g. {missing[:3]}")
    for n, shp in spec.items():
        a = W[n]
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{n}: shape {tuple(a.shape)} != {shp}")
        if a.dtype != np.float32:
            raise ValueError(f"{n}: dtype {a.dtype} != float32")


def save_npz(path: str, W: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k.replace("/", "|"): v for k, v in W.items()})


def load_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k.replace("|", "/"): z[k] for k in z.files}


def synthetic_text(hp: Hyperparams = _hp, B: int = 32, seed: int = 1234) -> np.ndarray:
    """Synthetic character batch shaped like ``data_load.load_data('synthesize')``
    (data_load.py:79-86): ids U{2..31}, one EOS (1) at the end, zero padded to max_N."""
    rng = np.random.default_rng(seed)
    L = np.zeros((B, hp.max_N), np.int32)
    for b in range(B):
        n = int(rng.integers(30, hp.max_N))          # total length incl. EOS, <= max_N - 1
        L[b, : n - 1] = rng.integers(2, len(hp.vocab), n - 1)
        L[b, n - 1] = 1
    return L
