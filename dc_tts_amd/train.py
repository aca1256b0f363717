"""The TRAINING path on the MI355X: what `train.py` obtains from TensorFlow's autodiff and optimizer, as calls
into libdctts_hip.so (include/dctts_train.h), the training graph built from them, and the loop of train.py's `__main__`.

  reference                                            here
  modules.py:143-197  hc(...) under tf.gradients       TrainOps.hc_forward / hc_backward
  modules.py:91-141   conv1d(...) under tf.gradients   TrainOps.conv1d_forward / conv1d_backward
  modules.py:199-247  conv1d_transpose(...)            TrainOps.conv1d_transpose_forward / _backward
  networks.py:126-155 Attention (training form)        TrainOps.attention_forward / attention_backward
  modules.py:13-42    embed                            TrainOps.embed_forward / embed_backward
  train.py:87,90,93-97  loss_mels, loss_bd1, loss_att  TrainOps.text2mel_losses
  train.py:104,107      loss_mags, loss_bd2            TrainOps.ssrn_losses
  train.py:119-131      clip_by_value + Adam           TrainOps.adam_step / adam_step_multi  (+ learning_rate_decay = utils.py:142-145)
  train.py:26-134       Graph(num, mode="train")       TrainGraph (train_op, save, restore)
  train.py:137-162      __main__ (Supervisor loop)     main  (`python -m dc_tts_amd.train <num>`; batches from data_load.get_batch)

There is no CPU or PyTorch fallback: without a GPU and the built library the constructors raise.
"""
import ctypes
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib
from .engine import DcttsError, _check, _ptr


def learning_rate_decay(init_lr: float, global_step: int, warmup_steps: float = 4000.0) -> float:
    """utils.py:142-145 (Noam scheme): init_lr * warmup^0.5 * min(step * warmup^-1.5, step^-0.5), step = global_step + 1."""
    step = float(global_step + 1)
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


_NAMES = {"HC": {"kernel": "/conv1d/kernel", "bias": "/conv1d/bias", "g1": "/H1/gamma", "b1": "/H1/beta", "g2": "/H2/gamma", "b2": "/H2/beta"},
          "C": {"kernel": "/conv1d/kernel", "bias": "/conv1d/bias", "gamma": "/normalize/gamma", "beta": "/normalize/beta"},
          "D": {"kernel": "/conv2d_transpose/kernel", "bias": "/conv2d_transpose/bias", "gamma": "/normalize/gamma", "beta": "/normalize/beta"}}


def layer_key(seed: int, step: int, prefix: str, index: int) -> int:
    """Dropout key of layer `index` of the network under scope `prefix` at training step `step`: forward and backward pass of a step
    derive the same key, different layers / steps / seeds different ones."""
    h = sum((i + 1) * ord(ch) for i, ch in enumerate(prefix)) % 65521
    return (int(seed) * 1000003 + int(step)) * 4294967296 + h * 65536 + int(index)


def network_forward(ops: "TrainOps", layers, W: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, padding: str, drop=None):
    """One network of networks.py on the trainer's TF-layout variables: returns (output, [input of every layer]) -- the activations
    network_backward needs.  x: the first layer's input (int32 character ids when that layer is the embedding).
    drop = (rate, seed, step): training=True, i.e. dropout behind every block but the embedding (modules.py:139,195,245)."""
    xs = []
    for li, L in enumerate(layers):
        sc = prefix + "/" + L.scope
        xs.append(x)
        if L.kind == "E":
            x = ops.embed_forward(x, W[sc + "/lookup_table"])
            continue
        p = {n: W[sc + suffix] for n, suffix in _NAMES[L.kind].items()}
        if L.kind == "HC":
            x = ops.hc_forward(x, p, rate=L.rate, padding=padding)
        elif L.kind == "D":
            x = ops.conv1d_transpose_forward(x, p)
        else:
            x = ops.conv1d_forward(x, p, rate=L.rate, padding=padding, act=None if L.act == "none" else L.act)
        if drop is not None:
            x = ops.dropout(x, layer_key(drop[1], drop[2], prefix, li), drop[0])
    return x, xs


def network_backward(ops: "TrainOps", layers, W: Dict[str, torch.Tensor], prefix: str, xs, dy: torch.Tensor, padding: str, drop=None):
    """Reverse pass over one network of networks.py given as its layer list (dc_tts_amd.layers.textenc_layers / audioenc_layers /
    audiodec_layers / ssrn_layers; prefix = its variable scope, e.g. "Text2Mel/AudioEnc"), W = the TF-named variables as device
    tensors, xs = the input of every layer (kept by the forward pass; character ids for the embedding), dy = gradient of the network's
    output.  Returns (gradient of the network's input, or None when the first layer is the embedding; {TF variable name: gradient}),
    i.e. what tf.gradients(loss, tf.trainable_variables(scope)) gives the optimizer at train.py:125."""
    grads, g = {}, dy
    layers = list(layers)
    for li, L, xin in zip(reversed(range(len(layers))), reversed(layers), reversed(list(xs))):
        sc = prefix + "/" + L.scope
        if drop is not None and L.kind != "E":
            g = ops.dropout(g, layer_key(drop[1], drop[2], prefix, li), drop[0])          # the mask of the forward pass, regenerated from its key
        if L.kind == "E":
            grads[sc + "/lookup_table"] = ops.embed_backward(xin, g, W[sc + "/lookup_table"].shape[0])
            g = None
            continue
        p = {n: W[sc + suffix] for n, suffix in _NAMES[L.kind].items()}
        if L.kind == "HC":
            r = ops.hc_backward(xin, g, p, rate=L.rate, padding=padding)
        elif L.kind == "D":
            r = ops.conv1d_transpose_backward(xin, g, p)
        else:
            r = ops.conv1d_backward(xin, g, p, rate=L.rate, padding=padding, act=None if L.act == "none" else L.act)
        for n, suffix in _NAMES[L.kind].items():
            grads[sc + suffix] = r[n]
        g = r["dx"]
    return g, grads


class TrainOps:
    """Device workspaces + the training-slice entry points for one GPU."""

    def __init__(self, device: int = None):
        if not torch.cuda.is_available():
            raise DcttsError("dc_tts_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        h = ctypes.c_void_p()
        self._ok(self.lib.dctts_train_create(ctypes.byref(h), self.device_index))
        self._h = h

    def _ok(self, rc: int):
        if rc != 0:
            raise DcttsError(f"libdctts_hip error {rc}: {_lib.last_error()}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dctts_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return int(self.lib.dctts_train_device_bytes(self._h))

    def tape(self, enable: bool):
        """Start of a training step: forward passes keep their pre-norm tensors, the reverse pass consumes them (dctts_train_tape)."""
        self._ok(self.lib.dctts_train_tape(self._h, 1 if enable else 0))

    # ------------------------------------------------------------------ forward passes on TF-layout variables
    def hc_forward(self, x, params, rate: int = 1, padding: str = "SAME") -> torch.Tensor:
        """y = hc(x) (modules.py:143-197) with params kernel (k, C, 2C), bias, g1, b1, g2, b2."""
        _check(x, "x", torch.float32, 3, self.device)
        B, T, C = x.shape
        y = torch.empty_like(x)
        self._ok(self.lib.dctts_train_hc_forward(self._h, _ptr(x), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["g1"]), _ptr(params["b1"]),
                                                 _ptr(params["g2"]), _ptr(params["b2"]), B, T, C, params["kernel"].shape[0], int(rate),
                                                 1 if padding.lower() == "causal" else 0, _ptr(y), self._stream()))
        return y

    def conv1d_forward(self, x, params, rate: int = 1, padding: str = "SAME", act: str = None) -> torch.Tensor:
        """y = conv1d(x) (modules.py:91-141) with params kernel (k, Cin, Cout), bias, gamma, beta."""
        _check(x, "x", torch.float32, 3, self.device)
        B, T, Cin = x.shape
        k, _, Cout = params["kernel"].shape
        y = torch.empty(B, T, Cout, dtype=torch.float32, device=self.device)
        self._ok(self.lib.dctts_train_conv1d_forward(self._h, _ptr(x), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["gamma"]), _ptr(params["beta"]),
                                                     B, T, Cin, Cout, k, int(rate), 1 if padding.lower() == "causal" else 0,
                                                     {None: 0, "relu": 1, "sigmoid": 2}[act], _ptr(y), self._stream()))
        return y

    def conv1d_transpose_forward(self, x, params) -> torch.Tensor:
        """y = conv1d_transpose(x) (modules.py:199-247) with params kernel (1, 3, Cout, Cin), bias, gamma, beta: (B, T, Cin) -> (B, 2T, Cout)."""
        _check(x, "x", torch.float32, 3, self.device)
        B, T, Cin = x.shape
        Cout = params["kernel"].shape[2]
        y = torch.empty(B, 2 * T, Cout, dtype=torch.float32, device=self.device)
        self._ok(self.lib.dctts_train_conv1d_transpose_forward(self._h, _ptr(x), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["gamma"]),
                                                               _ptr(params["beta"]), B, T, Cin, Cout, _ptr(y), self._stream()))
        return y

    def embed_forward(self, ids, table) -> torch.Tensor:
        """modules.py:13-42: (B, N) int32 ids -> (B, N, e); id 0 reads zeros."""
        _check(ids, "ids", torch.int32, 2, self.device); _check(table, "table", torch.float32, 2, self.device)
        y = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=self.device)
        self._ok(self.lib.dctts_train_embed_forward(self._h, _ptr(ids), _ptr(table), ids.numel(), table.shape[0], table.shape[1], _ptr(y), self._stream()))
        return y

    def attention_forward(self, Q, K, V) -> Tuple[torch.Tensor, torch.Tensor]:
        """Attention(Q, K, V) in its training form (networks.py:126-155, mononotic_attention=False): (R (B, T, 2d), alignments (B, N, T))."""
        for t, n in ((Q, "Q"), (K, "K"), (V, "V")):
            _check(t, n, torch.float32, 3, self.device)
        B, T, d = Q.shape
        N = K.shape[1]
        R = torch.empty(B, T, 2 * d, dtype=torch.float32, device=self.device); al = torch.empty(B, N, T, dtype=torch.float32, device=self.device)
        self._ok(self.lib.dctts_train_attention_forward(self._h, _ptr(Q), _ptr(K), _ptr(V), B, T, N, d, _ptr(R), _ptr(al), self._stream()))
        return R, al

    def dropout(self, x, key: int, rate: float) -> torch.Tensor:
        """tf.layers.dropout(rate, training=True): x * keep / (1 - rate) with the keep bits of `key`; on a gradient, the backward pass."""
        y = torch.empty_like(x)
        self._ok(self.lib.dctts_train_dropout(self._h, _ptr(x), _ptr(y), x.numel(), ctypes.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), ctypes.c_float(rate), self._stream()))
        return y

    def sigmoid(self, x) -> torch.Tensor:
        y = torch.empty_like(x)
        self._ok(self.lib.dctts_train_sigmoid(self._h, _ptr(x), _ptr(y), x.numel(), self._stream()))
        return y

    # ------------------------------------------------------------------ backward passes
    def hc_backward(self, x: torch.Tensor, dy: torch.Tensor, params: Dict[str, torch.Tensor], rate: int = 1,
                    padding: str = "SAME") -> Dict[str, torch.Tensor]:
        """Gradients of y = hc(x) (modules.py:143-197).  params: kernel (k, C, 2C), bias (2C), g1, b1, g2, b2 (C) -- the TF
        variables conv1d/kernel, conv1d/bias, H1/gamma, H1/beta, H2/gamma, H2/beta.  Returns dx and one gradient per parameter."""
        _check(x, "x", torch.float32, 3, self.device); _check(dy, "dy", torch.float32, 3, self.device)
        B, T, C = x.shape
        if tuple(dy.shape) != (B, T, C):
            raise ValueError(f"dy {tuple(dy.shape)} != x {tuple(x.shape)}")
        k = params["kernel"].shape[0]
        want = {"kernel": (k, C, 2 * C), "bias": (2 * C,), "g1": (C,), "b1": (C,), "g2": (C,), "b2": (C,)}
        for n, shp in want.items():
            _check(params[n], n, torch.float32, len(shp), self.device)
            if tuple(params[n].shape) != shp:
                raise ValueError(f"{n}: shape {tuple(params[n].shape)} != {shp}")
        if padding.lower() not in ("same", "causal"):
            raise ValueError(padding)
        out = {"dx": torch.empty_like(x)}
        for n in want:
            out[n] = torch.empty_like(params[n])
        self._ok(self.lib.dctts_train_hc_backward(
            self._h, _ptr(x), _ptr(dy), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["g1"]), _ptr(params["b1"]),
            _ptr(params["g2"]), _ptr(params["b2"]), B, T, C, k, int(rate), 1 if padding.lower() == "causal" else 0,
            _ptr(out["dx"]), _ptr(out["kernel"]), _ptr(out["bias"]), _ptr(out["g1"]), _ptr(out["b1"]), _ptr(out["g2"]), _ptr(out["b2"]),
            self._stream()))
        return out

    def conv1d_backward(self, x: torch.Tensor, dy: torch.Tensor, params: Dict[str, torch.Tensor], rate: int = 1, padding: str = "SAME",
                        act: str = None) -> Dict[str, torch.Tensor]:
        """Gradients of y = conv1d(x) (modules.py:91-141).  params: kernel (k, Cin, Cout), bias, gamma, beta (Cout) -- the TF
        variables conv1d/kernel, conv1d/bias, normalize/gamma, normalize/beta; act in (None, "relu", "sigmoid")."""
        _check(x, "x", torch.float32, 3, self.device); _check(dy, "dy", torch.float32, 3, self.device)
        B, T, Cin = x.shape
        k, _, Cout = params["kernel"].shape
        if tuple(dy.shape) != (B, T, Cout):
            raise ValueError(f"dy {tuple(dy.shape)} != {(B, T, Cout)}")
        want = {"kernel": (k, Cin, Cout), "bias": (Cout,), "gamma": (Cout,), "beta": (Cout,)}
        for n, shp in want.items():
            _check(params[n], n, torch.float32, len(shp), self.device)
            if tuple(params[n].shape) != shp:
                raise ValueError(f"{n}: shape {tuple(params[n].shape)} != {shp}")
        if padding.lower() not in ("same", "causal") or act not in (None, "relu", "sigmoid"):
            raise ValueError((padding, act))
        out = {"dx": torch.empty_like(x)}
        for n in want:
            out[n] = torch.empty_like(params[n])
        self._ok(self.lib.dctts_train_conv1d_backward(
            self._h, _ptr(x), _ptr(dy), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["gamma"]), _ptr(params["beta"]),
            B, T, Cin, Cout, k, int(rate), 1 if padding.lower() == "causal" else 0, {None: 0, "relu": 1, "sigmoid": 2}[act],
            _ptr(out["dx"]), _ptr(out["kernel"]), _ptr(out["bias"]), _ptr(out["gamma"]), _ptr(out["beta"]), self._stream()))
        return out

    def conv1d_transpose_backward(self, x: torch.Tensor, dy: torch.Tensor, params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Gradients of y = conv1d_transpose(x) (modules.py:199-247).  params: kernel (1, 3, Cout, Cin), bias, gamma, beta (Cout)."""
        _check(x, "x", torch.float32, 3, self.device); _check(dy, "dy", torch.float32, 3, self.device)
        B, T, Cin = x.shape
        Cout = params["kernel"].shape[2]
        want = {"kernel": (1, 3, Cout, Cin), "bias": (Cout,), "gamma": (Cout,), "beta": (Cout,)}
        if tuple(dy.shape) != (B, 2 * T, Cout):
            raise ValueError(f"dy {tuple(dy.shape)} != {(B, 2 * T, Cout)}")
        for n, shp in want.items():
            _check(params[n], n, torch.float32, len(shp), self.device)
            if tuple(params[n].shape) != shp:
                raise ValueError(f"{n}: shape {tuple(params[n].shape)} != {shp}")
        out = {"dx": torch.empty_like(x)}
        for n in want:
            out[n] = torch.empty_like(params[n])
        self._ok(self.lib.dctts_train_conv1d_transpose_backward(
            self._h, _ptr(x), _ptr(dy), _ptr(params["kernel"]), _ptr(params["bias"]), _ptr(params["gamma"]), _ptr(params["beta"]), B, T, Cin, Cout,
            _ptr(out["dx"]), _ptr(out["kernel"]), _ptr(out["bias"]), _ptr(out["gamma"]), _ptr(out["beta"]), self._stream()))
        return out

    def attention_backward(self, Q, K, V, dR, dalignments) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Gradients (dQ, dK, dV) of Attention(Q, K, V) in its training form (networks.py:126-155, mononotic_attention=False), given
        dR (B, T, 2d) and the gradient with respect to the returned alignments (B, N, T)."""
        for t, n in ((Q, "Q"), (K, "K"), (V, "V"), (dR, "dR"), (dalignments, "dalignments")):
            _check(t, n, torch.float32, 3, self.device)
        B, T, d = Q.shape
        N = K.shape[1]
        if tuple(K.shape) != (B, N, d) or tuple(V.shape) != (B, N, d) or tuple(dR.shape) != (B, T, 2 * d) or tuple(dalignments.shape) != (B, N, T):
            raise ValueError("attention_backward: shapes do not agree")
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        self._ok(self.lib.dctts_train_attention_backward(self._h, _ptr(Q), _ptr(K), _ptr(V), _ptr(dR), _ptr(dalignments), B, T, N, d,
                                                         _ptr(dQ), _ptr(dK), _ptr(dV), self._stream()))
        return dQ, dK, dV

    def embed_backward(self, ids: torch.Tensor, dy: torch.Tensor, vocab: int) -> torch.Tensor:
        """Gradient of the lookup table (modules.py:13-42; row 0 receives none).  ids (B, N) int32, dy (B, N, e)."""
        _check(ids, "ids", torch.int32, 2, self.device); _check(dy, "dy", torch.float32, 3, self.device)
        if tuple(dy.shape[:2]) != tuple(ids.shape):
            raise ValueError("embed_backward: shapes do not agree")
        dT = torch.empty(vocab, dy.shape[2], dtype=torch.float32, device=self.device)
        self._ok(self.lib.dctts_train_embed_backward(self._h, _ptr(ids), _ptr(dy), ids.numel(), int(vocab), dy.shape[2], _ptr(dT), self._stream()))
        return dT

    def text2mel_losses(self, Y, Y_logits, mels, alignments, max_N: int, max_T: int):
        """train.py:85-100: returns (losses (3,) = loss_mels, loss_bd1, loss_att; dY, dY_logits, dalignments)."""
        for t, n in ((Y, "Y"), (Y_logits, "Y_logits"), (mels, "mels"), (alignments, "alignments")):
            _check(t, n, torch.float32, 3, self.device)
        B, T, M = Y.shape
        if tuple(Y_logits.shape) != (B, T, M) or tuple(mels.shape) != (B, T, M) or alignments.shape[0] != B or alignments.shape[2] != T:
            raise ValueError("text2mel_losses: shapes do not agree")
        N = alignments.shape[1]
        losses = torch.empty(3, dtype=torch.float32, device=self.device)
        dY, dlog, dA = torch.empty_like(Y), torch.empty_like(Y_logits), torch.empty_like(alignments)
        self._ok(self.lib.dctts_train_text2mel_losses(self._h, _ptr(Y), _ptr(Y_logits), _ptr(mels), _ptr(alignments), B, T, M, N,
                                                      int(max_N), int(max_T), _ptr(losses), _ptr(dY), _ptr(dlog), _ptr(dA), self._stream()))
        return losses, dY, dlog, dA

    def ssrn_losses(self, Z, Z_logits, mags):
        """train.py:102-110: returns (losses (2,) = loss_mags, loss_bd2; dZ, dZ_logits)."""
        for t, n in ((Z, "Z"), (Z_logits, "Z_logits"), (mags, "mags")):
            _check(t, n, torch.float32, Z.dim(), self.device)
        if Z_logits.shape != Z.shape or mags.shape != Z.shape:
            raise ValueError("ssrn_losses: shapes do not agree")
        losses = torch.empty(2, dtype=torch.float32, device=self.device)
        dZ, dlog = torch.empty_like(Z), torch.empty_like(Z_logits)
        self._ok(self.lib.dctts_train_ssrn_losses(self._h, _ptr(Z), _ptr(Z_logits), _ptr(mags), Z.numel(), _ptr(losses), _ptr(dZ), _ptr(dlog), self._stream()))
        return losses, dZ, dlog

    def adam_step(self, var, grad, m, v, step: int, lr: float):
        """train.py:119-131 in place on var / m / v: clip_by_value(grad, -1, 1), then tf.train.AdamOptimizer's update; step is 1-based."""
        for t, n in ((var, "var"), (grad, "grad"), (m, "m"), (v, "v")):
            _check(t, n, torch.float32, var.dim(), self.device)
        self._ok(self.lib.dctts_train_adam_step(self._h, _ptr(var), _ptr(grad), _ptr(m), _ptr(v), var.numel(), int(step), float(lr), self._stream()))


    def adam_step_multi(self, vars_, grads, ms, vs, step: int, lr: float):
        """The same update on lists of variables in ONE launch (train.py:131 applies the whole gradient list); same results as a loop of
        adam_step, without ~200 launches and calls per step."""
        import ctypes
        n = len(vars_)
        if not (n == len(grads) == len(ms) == len(vs)) or n == 0:
            raise ValueError("adam_step_multi: the four lists must have the same non-zero length")
        for i in range(n):
            for t, nm in ((vars_[i], "var"), (grads[i], "grad"), (ms[i], "m"), (vs[i], "v")):
                _check(t, nm, torch.float32, vars_[i].dim(), self.device)
                if t.numel() != vars_[i].numel():
                    raise ValueError("adam_step_multi: %s[%d] has %d elements, its variable %d" % (nm, i, t.numel(), vars_[i].numel()))
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        ns = (ctypes.c_longlong * n)(*[t.numel() for t in vars_])
        self._ok(self.lib.dctts_train_adam_step_multi(self._h, n, arr(vars_), arr(grads), arr(ms), arr(vs), ns, int(step), float(lr), self._stream()))


class TrainGraph:
    """train.py:26-134 for mode == "train", one network at a time as the reference does: num = 1 trains Text2Mel, num = 2 trains SSRN.
    Holds the variables of that network in TF layout on the device, their Adam moments and `global_step`; `train_op(...)` is one
    `sess.run(g.train_op)`: forward (keeping every layer's input), the losses of train.py:85-110, the gradient of every variable,
    clip_by_value(-1, 1) + Adam with the Noam learning rate (train.py:116-131).  training=True applies dropout (hp.dropout_rate behind
    every block, modules.py:139,195,245) from a counter-based hash of (seed, global_step, layer) -- TensorFlow's random stream cannot be
    reproduced.  The loop around it (train.py:137-162: batches from data_load.get_batch, a checkpoint every 1000 steps, resume from
    the newest one) is `main` below; TensorBoard summaries (train.py:112-113,133) are not written."""

    def __init__(self, num: int, weights, hp, device: int = None, training: bool = True, seed: int = 0):
        from .layers import audiodec_layers, audioenc_layers, ssrn_layers, textenc_layers
        if num not in (1, 2):
            raise ValueError("num: 1 = Text2Mel, 2 = SSRN (train.py:141)")
        self.num, self.hp = num, hp
        self.training, self.seed = bool(training), int(seed)       # training=True: dropout hp.dropout_rate behind every block (train.py:55-72)
        self.ops = TrainOps(device)
        prefix = "Text2Mel/" if num == 1 else "SSRN/"
        import numpy as _np
        self.W = {n: torch.from_numpy(_np.ascontiguousarray(v, dtype=_np.float32)).to(self.ops.device) for n, v in weights.items() if n.startswith(prefix)}
        self.m = {n: torch.zeros_like(v) for n, v in self.W.items()}
        self.v = {n: torch.zeros_like(v) for n, v in self.W.items()}
        self.global_step = 0
        self.alignments = None
        self._te, self._ae, self._ad, self._ss = textenc_layers(hp), audioenc_layers(hp), audiodec_layers(hp), ssrn_layers(hp)

    def loss_and_grads(self, *batch):
        """num == 1: (L (B, N) int32, mels (B, T, n_mels));  num == 2: (mels (B, T, n_mels), mags (B, 4T, n_linear)).
        Returns (losses on the device: loss_mels, loss_bd1, loss_att / loss_mags, loss_bd2;  {TF variable name: gradient})."""
        ops, W, hp = self.ops, self.W, self.hp
        grads = {}
        ops.tape(True)                                          # the reverse pass reuses the forward pass's pre-norm tensors
        drop = (hp.dropout_rate, self.seed, self.global_step) if self.training and hp.dropout_rate > 0 else None
        if self.num == 1:
            L, mels = batch
            d = hp.d
            S = torch.cat((torch.zeros_like(mels[:, :1]), mels[:, :-1]), 1).contiguous()              # train.py:51
            KV, xs_te = network_forward(ops, self._te, W, "Text2Mel/TextEnc", L, "SAME", drop)
            K, V = KV[..., :d].contiguous(), KV[..., d:].contiguous()                                   # networks.py:69
            Q, xs_ae = network_forward(ops, self._ae, W, "Text2Mel/AudioEnc", S, "CAUSAL", drop)
            R, al = ops.attention_forward(Q, K, V)
            logits, xs_ad = network_forward(ops, self._ad, W, "Text2Mel/AudioDec", R, "CAUSAL", drop)
            Y = ops.sigmoid(logits)
            self.alignments = al                                                                         # (B, N, T), train.py:160 plots [0]
            losses, dY, dlog, dA = ops.text2mel_losses(Y, logits, mels, al, hp.max_N, hp.max_T)
            dlog = dlog + dY * Y * (1.0 - Y)                                                            # Y = sigmoid(Y_logits) (networks.py:210)
            dR, g = network_backward(ops, self._ad, W, "Text2Mel/AudioDec", xs_ad, dlog, "CAUSAL", drop); grads.update(g)
            dQ, dK, dV = ops.attention_backward(Q, K, V, dR, dA)
            _, g = network_backward(ops, self._ae, W, "Text2Mel/AudioEnc", xs_ae, dQ, "CAUSAL", drop); grads.update(g)
            _, g = network_backward(ops, self._te, W, "Text2Mel/TextEnc", xs_te, torch.cat((dK, dV), -1).contiguous(), "SAME", drop); grads.update(g)
        else:
            mels, mags = batch
            logits, xs = network_forward(ops, self._ss, W, "SSRN", mels, "SAME", drop)
            Z = ops.sigmoid(logits)
            losses, dZ, dlog = ops.ssrn_losses(Z, logits, mags)
            _, grads = network_backward(ops, self._ss, W, "SSRN", xs, dlog + dZ * Z * (1.0 - Z), "SAME", drop)
        return losses, grads

    def train_op(self, *batch):
        """One training step; returns the losses (device tensor) of the step, evaluated before the update as sess.run does."""
        losses, grads = self.loss_and_grads(*batch)
        lr = learning_rate_decay(self.hp.lr, self.global_step)                                          # train.py:116
        names = list(grads)
        self.ops.adam_step_multi([self.W[n] for n in names], [grads[n] for n in names], [self.m[n] for n in names], [self.v[n] for n in names],
                                 self.global_step + 1, lr)
        self.global_step += 1
        return losses

    def save(self, logdir: str) -> str:
        """train.py:158: the network's variables, Adam slots and global_step as a TF V2 checkpoint under `logdir` (the reference uses
        `<hp.logdir>-1` for Text2Mel and `-2` for SSRN); dc_tts_amd.tf_checkpoint.load_reference_weights reads it back."""
        from .tf_checkpoint import save_checkpoint
        torch.cuda.synchronize(self.ops.device)
        cpu = lambda d: {n: t.cpu().numpy() for n, t in d.items()}
        return save_checkpoint(logdir, cpu(self.W), self.global_step, {"Adam": cpu(self.m), "Adam_1": cpu(self.v)})

    def restore(self, logdir: str) -> bool:
        """What tf.train.Supervisor does when `logdir` holds a checkpoint (train.py:146-147): variables, Adam slots and global_step
        come from the newest one and training continues from there.  Returns False (state untouched) when there is none."""
        import os
        from .tf_checkpoint import latest_checkpoint, read_checkpoint
        if not os.path.exists(os.path.join(logdir, "checkpoint")):
            return False
        prefix = latest_checkpoint(logdir)
        names = list(self.W)
        want = names + [n + "/Adam" for n in names] + [n + "/Adam_1" for n in names] + ["gs/global_step"]
        t = read_checkpoint(prefix, want)
        for n in names:
            for dst, key in ((self.W, n), (self.m, n + "/Adam"), (self.v, n + "/Adam_1")):
                a = np.array(t[key], dtype=np.float32)
                if tuple(a.shape) != tuple(dst[n].shape):
                    raise ValueError("%s: checkpoint shape %s, model shape %s" % (key, a.shape, tuple(dst[n].shape)))
                dst[n].copy_(torch.from_numpy(a))
        self.global_step = int(t["gs/global_step"])
        return True


def plot_alignment(alignment, gs: str, dir: str) -> str:
    """utils.py:116-132: `alignment` (encoder_steps, decoder_steps) as `<dir>/alignment_<gs>.png` (matplotlib, when installed;
    otherwise the array itself as .npy)."""
    import os
    os.makedirs(dir, exist_ok=True)
    try:
        import matplotlib
        matplotlib.use("pdf")
        import matplotlib.pyplot as plt
    except ImportError:
        path = "{}/alignment_{}.npy".format(dir, gs)
        np.save(path, np.asarray(alignment))
        return path
    fig, ax = plt.subplots()
    im = ax.imshow(alignment)
    fig.colorbar(im)
    plt.title("{} Steps".format(gs))
    path = "{}/alignment_{}.png".format(dir, gs)
    plt.savefig(path, format="png")
    plt.close(fig)
    return path


def initial_variables(hp, seed: int = 0):
    """Where a from-scratch run of train.py starts: `synthetic_weights(..., perturb=False)`."""
    from .weights import synthetic_weights
    return synthetic_weights(hp, seed=seed, perturb=False)


def main(argv=None, hp=None, save_every: int = 1000, batches=None, init_seed: int = 0) -> int:
    """`python -m dc_tts_amd.train <num>` = train.py:137-162.  num: 1 trains Text2Mel, 2 trains SSRN, each into `<hp.logdir>-<num>`.
    Variables start from the reference's initialisers (dc_tts_amd.weights.synthetic_weights) or, when the log directory holds a
    checkpoint, from the newest one (Supervisor).  Batches come from data_load.get_batch (bucketed by text length, B = hp.B, arrays
    written by `python -m dc_tts_amd.prepo` when hp.prepro); every `save_every` steps: a checkpoint `model_gs_<k>k` and, for Text2Mel,
    the first utterance's alignment plot; stops after hp.num_iterations steps.  `batches`: an iterable to use instead (tests)."""
    import argparse
    import sys
    from .hyperparams import hp as _hp
    from .weights import synthetic_weights
    hp = hp or _hp
    ap = argparse.ArgumentParser(description="DC-TTS training on MI355X: 1 = Text2Mel, 2 = SSRN (train.py)")
    ap.add_argument("num", type=int, choices=(1, 2))
    ap.add_argument("--data", default=None, help="corpus directory holding transcript.csv (hp.data)")
    ap.add_argument("--prepro-dir", default=".", help="directory holding mels/ and mags/ written by `python -m dc_tts_amd.prepo`")
    ap.add_argument("--logdir", default=None, help="hp.logdir; '-<num>' is appended")
    ap.add_argument("--num-iterations", type=int, default=None)
    args = ap.parse_args(argv)
    if args.data: hp = hp.replace(data=args.data)
    if args.logdir: hp = hp.replace(logdir=args.logdir)
    if args.num_iterations is not None: hp = hp.replace(num_iterations=args.num_iterations)
    num = args.num
    # the reference's initialisers (modules.py / tf.layers defaults): variance-scaling truncated-normal kernels, truncated normal 0.1 for the
    # embedding, layer-norm gamma = 1, beta = 0, conv bias = 0 -- perturb=False; the perturbed variant is for parity tests only
    g = TrainGraph(num, initial_variables(hp, init_seed), hp)
    logdir = hp.logdir + "-" + str(num)
    resumed = g.restore(logdir)
    print("Training Graph loaded" + (" (resumed at global_step %d)" % g.global_step if resumed else ""), file=sys.stderr)
    if batches is None:
        from .data_load import get_batch
        batches = get_batch(hp, seed=g.global_step, prepro_dir=args.prepro_dir)                    # texts padded to the batch's longest, as data_load.py:152-160 does
    dev = g.ops.device
    for texts, mels, mags, _ in batches:
        if num == 1:
            batch = (torch.from_numpy(texts).to(dev), torch.from_numpy(mels).to(dev))
        else:
            batch = (torch.from_numpy(mels).to(dev), torch.from_numpy(mags).to(dev))
        g.train_op(*batch)
        gs = g.global_step
        if gs % save_every == 0:                                                                    # train.py:157-164
            g.save(logdir)
            if num == 1:
                plot_alignment(g.alignments[0].cpu().numpy(), str(gs // 1000).zfill(3) + "k", logdir)
        if gs > hp.num_iterations:                                                                  # train.py:167
            break
    print("Done", file=sys.stderr)
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
