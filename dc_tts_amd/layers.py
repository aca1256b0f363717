"""Layer tables of the four conv stacks on the synthesis path.

Each table lists, in source order, the layers that ``networks.py`` builds, with
the variable-scope names the reference derives from its running index ``i`` so
that weight names equal TF checkpoint names:

  TextEnc   networks.py:14-71    scope Text2Mel/TextEnc   padding SAME
  AudioEnc  networks.py:73-124   scope Text2Mel/AudioEnc  padding CAUSAL
  AudioDec  networks.py:157-212  scope Text2Mel/AudioDec  padding CAUSAL
  SSRN      networks.py:214-292  scope SSRN               padding SAME

Layer kinds: ``E`` embed (modules.py:13), ``C`` conv1d (modules.py:91),
``HC`` highway conv (modules.py:143), ``D`` conv1d_transpose (modules.py:199).
"""
from dataclasses import dataclass
from typing import Dict, List, Tuple

from .hyperparams import Hyperparams


@dataclass(frozen=True)
class Layer:
    scope: str          # e.g. "HC_7"
    kind: str           # "E" | "C" | "HC" | "D"
    cin: int
    cout: int           # channels leaving the layer (C for HC; conv itself emits 2C)
    size: int = 1       # kernel taps
    rate: int = 1       # dilation
    act: str = "none"   # "none" | "relu" (sigmoid after the last layer is applied by the network)

    @property
    def conv_filters(self) -> int:
        return 2 * self.cout if self.kind == "HC" else self.cout


def textenc_layers(hp: Hyperparams) -> List[Layer]:
    L = [Layer("embed_1", "E", len(hp.vocab), hp.e)]
    L.append(Layer("C_2", "C", hp.e, 2 * hp.d, act="relu"))
    L.append(Layer("C_3", "C", 2 * hp.d, 2 * hp.d))
    i = 4
    for _ in range(2):
        for j in range(4):
            L.append(Layer(f"HC_{i}", "HC", 2 * hp.d, 2 * hp.d, 3, 3 ** j)); i += 1
    for _ in range(2):
        L.append(Layer(f"HC_{i}", "HC", 2 * hp.d, 2 * hp.d, 3, 1)); i += 1
    for _ in range(2):
        L.append(Layer(f"HC_{i}", "HC", 2 * hp.d, 2 * hp.d, 1, 1)); i += 1
    return L


def audioenc_layers(hp: Hyperparams) -> List[Layer]:
    L = [Layer("C_1", "C", hp.n_mels, hp.d, act="relu"),
         Layer("C_2", "C", hp.d, hp.d, act="relu"),
         Layer("C_3", "C", hp.d, hp.d)]
    i = 4
    for _ in range(2):
        for j in range(4):
            L.append(Layer(f"HC_{i}", "HC", hp.d, hp.d, 3, 3 ** j)); i += 1
    for _ in range(2):
        L.append(Layer(f"HC_{i}", "HC", hp.d, hp.d, 3, 3)); i += 1
    return L


def audiodec_layers(hp: Hyperparams) -> List[Layer]:
    L = [Layer("C_1", "C", 2 * hp.d, hp.d)]
    i = 2
    for j in range(4):
        L.append(Layer(f"HC_{i}", "HC", hp.d, hp.d, 3, 3 ** j)); i += 1
    for _ in range(2):
        L.append(Layer(f"HC_{i}", "HC", hp.d, hp.d, 3, 1)); i += 1
    for _ in range(3):
        L.append(Layer(f"C_{i}", "C", hp.d, hp.d, act="relu")); i += 1
    L.append(Layer(f"C_{i}", "C", hp.d, hp.n_mels))
    return L


def ssrn_layers(hp: Hyperparams) -> List[Layer]:
    c = hp.c
    L = [Layer("C_1", "C", hp.n_mels, c)]
    i = 2
    for j in range(2):
        L.append(Layer(f"HC_{i}", "HC", c, c, 3, 3 ** j)); i += 1
    for _ in range(2):
        L.append(Layer(f"D_{i}", "D", c, c, 3, 1)); i += 1
        for j in range(2):
            L.append(Layer(f"HC_{i}", "HC", c, c, 3, 3 ** j)); i += 1
    L.append(Layer(f"C_{i}", "C", c, 2 * c)); i += 1
    for _ in range(2):
        L.append(Layer(f"HC_{i}", "HC", 2 * c, 2 * c, 3, 1)); i += 1
    L.append(Layer(f"C_{i}", "C", 2 * c, hp.n_linear)); i += 1
    for _ in range(2):
        L.append(Layer(f"C_{i}", "C", hp.n_linear, hp.n_linear, act="relu")); i += 1
    L.append(Layer(f"C_{i}", "C", hp.n_linear, hp.n_linear))
    return L


NETS = {
    "Text2Mel/TextEnc": textenc_layers,
    "Text2Mel/AudioEnc": audioenc_layers,
    "Text2Mel/AudioDec": audiodec_layers,
    "SSRN": ssrn_layers,
}


def variable_shapes(hp: Hyperparams) -> Dict[str, Tuple[int, ...]]:
    """TF variable name -> shape for every trainable variable the synthesis graph
    restores (synthesize.py:32-40; names from tf.layers / contrib defaults)."""
    out: Dict[str, Tuple[int, ...]] = {}
    for net, fn in NETS.items():
        for l in fn(hp):
            p = f"{net}/{l.scope}"
            if l.kind == "E":
                out[f"{p}/lookup_table"] = (l.cin, l.cout)
            elif l.kind == "C":
                out[f"{p}/conv1d/kernel"] = (l.size, l.cin, l.cout)
                out[f"{p}/conv1d/bias"] = (l.cout,)
                out[f"{p}/normalize/beta"] = (l.cout,)
                out[f"{p}/normalize/gamma"] = (l.cout,)
            elif l.kind == "HC":
                out[f"{p}/conv1d/kernel"] = (l.size, l.cin, 2 * l.cout)
                out[f"{p}/conv1d/bias"] = (2 * l.cout,)
                for h in ("H1", "H2"):
                    out[f"{p}/{h}/beta"] = (l.cout,)
                    out[f"{p}/{h}/gamma"] = (l.cout,)
            elif l.kind == "D":
                out[f"{p}/conv2d_transpose/kernel"] = (1, l.size, l.cout, l.cin)
                out[f"{p}/conv2d_transpose/bias"] = (l.cout,)
                out[f"{p}/normalize/beta"] = (l.cout,)
                out[f"{p}/normalize/gamma"] = (l.cout,)
    return out


def audiodec_cone(hp: Hyperparams) -> List[List[int]]:
    """Row offsets (relative to the step's newest frame j, all <= 0) that each
    AudioDec layer must emit so that its last layer can emit row j exactly
    (the dependency cone 85/83/45/15/5/3/1).

    Returns one sorted (descending, starting at 0) list per layer, aligned with
    ``audiodec_layers(hp)``; entry ``l`` is the set of OUTPUT rows of layer ``l``.
    """
    layers = audiodec_layers(hp)
    need = {0}
    out = [None] * len(layers)
    for idx in range(len(layers) - 1, -1, -1):
        l = layers[idx]
        out[idx] = sorted(need, reverse=True)
        if l.size > 1:
            nxt = set()
            for o in need:
                for j in range(l.size):
                    nxt.add(o - j * l.rate)
            need = nxt
    return out
