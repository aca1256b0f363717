"""CPU ORACLE for the TRAINING path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

What it restates, in float64 numpy with every derivative written out by hand:
  * the backward pass of the highway-convolution block ``hc`` (modules.py:143-197): conv to 2C -> split ->
    layer-norm(H1), layer-norm(H2) (modules.py:45-64, eps 1e-12) -> sigmoid gate -> highway mix, and of ``conv1d``
    (modules.py:91-141): conv -> layer-norm -> activation;
  * the three Text2Mel losses and the two SSRN losses of train.py:85-113 (L1, binary divergence =
    ``sigmoid_cross_entropy_with_logits``, guided attention with ``utils.py:134-140``'s weight matrix) and their
    gradients with respect to the network outputs;
  * the Noam learning-rate schedule (utils.py:142-145) and the clip / Adam step of train.py:119-131.

PARITY UNPINNED against TensorFlow (the forward restatement in oracle/dctts_ref.py is pinned to the reference's own Python since round 4, its
derivatives are not): TensorFlow is not installable here, so nothing ties these derivatives to
``tf.gradients``.  They are pinned by mathematics instead: ``tests/test_train_oracle.py`` checks every gradient
against central finite differences of the float64 forward pass (which IS oracle/dctts_ref.py's forward) and against
torch.autograd on the same blocks written with torch.nn.functional.
"""
import numpy as np

from oracle import dctts_ref as O

LN_EPS = O.LN_EPS


# ----------------------------------------------------------------------------- modules.py, backward
def conv_pads(k, rate, padding):
    """Left / right zero padding of tf.layers.conv1d as called at modules.py:134,187 (see dctts_ref._conv)."""
    total = (k - 1) * rate
    if padding.lower() == "causal":
        return total, 0
    pl = total // 2
    return pl, total - pl


def conv_bwd(x, W, dy, rate, padding):
    """Gradients of y = conv(x, W) + b (dctts_ref._conv):  (dx, dW, db)."""
    k = W.shape[0]
    pl, pr = conv_pads(k, rate, padding)
    T = x.shape[1]
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    dW = np.zeros_like(W)
    for j in range(k):
        xs = xp[:, j * rate: j * rate + T, :]
        dW[j] = np.einsum("bti,bto->io", xs, dy)
        dxp[:, j * rate: j * rate + T, :] += dy @ W[j].T
    return dxp[:, pl: pl + T, :], dW, dy.sum(axis=(0, 1))


def normalize_bwd(x, gamma, dy):
    """Gradients of y = normalize(x, gamma, beta) (modules.py:45-64; biased variance, eps inside the root):  (dx, dgamma, dbeta)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xh = (x - mean) * rstd
    dxh = dy * gamma
    dx = rstd * (dxh - dxh.mean(axis=-1, keepdims=True) - xh * (dxh * xh).mean(axis=-1, keepdims=True))
    red = tuple(range(x.ndim - 1))
    return dx, (dy * xh).sum(axis=red), dy.sum(axis=red)


def c_fwd(x, p, rate, padding, act):
    """modules.py:91-141 with the parameters as a dict: kernel (k, Cin, Cout), bias, gamma, beta; act in (None, "relu", "sigmoid")."""
    P = {"s/conv1d/kernel": p["kernel"], "s/conv1d/bias": p["bias"], "s/normalize/gamma": p["gamma"], "s/normalize/beta": p["beta"]}
    return O.conv1d(x, P, "s", rate=rate, padding=padding, act={None: None, "relu": O.relu, "sigmoid": O.sigmoid}[act])


def c_bwd(x, p, dy, rate, padding, act):
    """Backward of conv1d (conv -> layer-norm -> activation).  Returns dict(dx, kernel, bias, gamma, beta)."""
    H = O._conv(x, p["kernel"], p["bias"], rate, padding)
    n = O.normalize(H, p["gamma"], p["beta"])
    if act == "relu":
        dn = dy * (n > 0)
    elif act == "sigmoid":
        y = O.sigmoid(n)
        dn = dy * y * (1.0 - y)
    else:
        dn = dy
    dH, dg, db = normalize_bwd(H, p["gamma"], dn)
    dx, dW, dbias = conv_bwd(x, p["kernel"], dH, rate, padding)
    return {"dx": dx, "kernel": dW, "bias": dbias, "gamma": dg, "beta": db}


def hc_fwd(x, p, rate, padding):
    """modules.py:143-197 with the parameters as a dict: kernel (k, C, 2C), bias, g1, b1, g2, b2."""
    P = {"s/conv1d/kernel": p["kernel"], "s/conv1d/bias": p["bias"], "s/H1/gamma": p["g1"], "s/H1/beta": p["b1"],
         "s/H2/gamma": p["g2"], "s/H2/beta": p["b2"]}
    return O.hc(x, P, "s", rate=rate, padding=padding)


def hc_bwd(x, p, dy, rate, padding):
    """Backward of hc.  Returns dict(dx, kernel, bias, g1, b1, g2, b2) of gradients."""
    H = O._conv(x, p["kernel"], p["bias"], rate, padding)
    C = H.shape[-1] // 2
    H1, H2 = H[..., :C], H[..., C:]
    n1 = O.normalize(H1, p["g1"], p["b1"])
    n2 = O.normalize(H2, p["g2"], p["b2"])
    s = O.sigmoid(n1)
    # y = s * n2 + (1 - s) * x
    dn2 = dy * s
    ds = dy * (n2 - x)
    dx_direct = dy * (1.0 - s)
    dn1 = ds * s * (1.0 - s)
    dH1, dg1, db1 = normalize_bwd(H1, p["g1"], dn1)
    dH2, dg2, db2 = normalize_bwd(H2, p["g2"], dn2)
    dH = np.concatenate([dH1, dH2], axis=-1)
    dx_conv, dW, dbias = conv_bwd(x, p["kernel"], dH, rate, padding)
    return {"dx": dx_direct + dx_conv, "kernel": dW, "bias": dbias, "g1": dg1, "b1": db1, "g2": dg2, "b2": db2, "dH": dH}


def d_fwd(x, p):
    """modules.py:199-247 with the parameters as a dict: kernel (1, 3, Cout, Cin), bias, gamma, beta."""
    P = {"s/conv2d_transpose/kernel": p["kernel"], "s/conv2d_transpose/bias": p["bias"], "s/normalize/gamma": p["gamma"], "s/normalize/beta": p["beta"]}
    return O.conv1d_transpose(x, P, "s")


def d_bwd(x, p, dy):
    """Backward of conv1d_transpose (out[2t] = b + x[t] W0^T + x[t-1] W2^T, out[2t+1] = b + x[t] W1^T, then layer-norm).
    Returns dict(dx, kernel (1, 3, Cout, Cin), bias, gamma, beta)."""
    W = p["kernel"][0]
    B, T, _ = x.shape
    xm1 = np.pad(x, ((0, 0), (1, 0), (0, 0)))[:, :T, :]
    H = np.zeros((B, 2 * T, W.shape[1]), x.dtype)
    H[:, 0::2, :] = x @ W[0].T + xm1 @ W[2].T
    H[:, 1::2, :] = x @ W[1].T
    H = H + p["bias"]
    dH, dg, db = normalize_bwd(H, p["gamma"], dy)
    de, do = dH[:, 0::2, :], dH[:, 1::2, :]
    dW = np.zeros_like(W)
    dW[0] = np.einsum("bto,bti->oi", de, x)
    dW[1] = np.einsum("bto,bti->oi", do, x)
    dW[2] = np.einsum("bto,bti->oi", de, xm1)
    dx = de @ W[0] + do @ W[1]
    dx[:, :-1, :] += de[:, 1:, :] @ W[2]                  # x[t] also feeds out[2(t+1)] through the t-1 tap
    return {"dx": dx, "kernel": dW[None], "bias": dH.sum(axis=(0, 1)), "gamma": dg, "beta": db}


def attention_bwd(Q, K, V, dR, dAl, d):
    """Backward of the training-time Attention (networks.py:126-155 with mononotic_attention=False): A = softmax(Q K^T / sqrt(d)),
    R = [A V ; Q], alignments = A^T.  dR (B, T, 2d), dAl (B, N, T) = gradient with respect to the returned alignments.
    Returns (dQ, dK, dV)."""
    scale = 1.0 / np.sqrt(float(d))
    S = (Q @ K.transpose(0, 2, 1)) * scale
    A = np.exp(S - S.max(axis=-1, keepdims=True)); A /= A.sum(axis=-1, keepdims=True)
    dRA = dR[..., :d]
    dA = dRA @ V.transpose(0, 2, 1) + dAl.transpose(0, 2, 1)
    dV = A.transpose(0, 2, 1) @ dRA
    dS = A * (dA - (A * dA).sum(axis=-1, keepdims=True)) * scale
    return dS @ K + dR[..., d:], dS.transpose(0, 2, 1) @ Q, dV


def embed_bwd(ids, dy, vocab):
    """Backward of embed (modules.py:13-42): row 0 of the table is replaced by zeros at lookup time (:36-38), so it receives none."""
    dT = np.zeros((vocab, dy.shape[-1]), dy.dtype)
    np.add.at(dT, ids.reshape(-1), dy.reshape(-1, dy.shape[-1]))
    dT[0] = 0
    return dT


def dropout_mask(key, n, rate):
    """Keep mask of tf.layers.dropout as THIS implementation draws it (csrc/train_kernels.h: dropout_kernel): element i is kept iff the
    top 24 bits of splitmix64(splitmix64(key) + i), as a fraction, are >= rate (the key is hashed first: layer keys differ only in their low
    bits, and splitmix64(key + i) would make consecutive layers' masks shifted copies of one stream).  (TensorFlow's random stream cannot be
    reproduced.)"""
    def mix(z):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        k = mix(np.array([int(key) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        z = mix(k + np.arange(n, dtype=np.uint64))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(rate)


def dropout(x, key, rate):
    """y = x * keep / (1 - rate) (modules.py:139,195,245 with training=True); the same call on dy is the backward pass."""
    keep = dropout_mask(key, x.size, rate).reshape(x.shape)
    return np.where(keep, x * (1.0 / (1.0 - rate)), 0.0)


def layer_key(seed, step, prefix, index):
    """The dropout key of layer `index` of network `prefix` at training step `step` (dc_tts_amd/train.py uses the same rule)."""
    h = sum((i + 1) * ord(ch) for i, ch in enumerate(prefix)) % 65521
    return (int(seed) * 1000003 + int(step)) * 4294967296 + h * 65536 + int(index)


# ----------------------------------------------------------------------------- networks.py, the four stacks layer by layer
# Spelled out here, line by line from networks.py, NOT imported from dc_tts_amd.layers: the oracle and the product must not share the
# structure one is supposed to check in the other.  (scope, kind, taps, dilation, activation); LJ hyper-parameters (hyperparams.py:7-47)
# only enter through the variable shapes.  The scope index is the reference's running counter `i`.
class _L:
    def __init__(self, scope, kind, size=1, rate=1, act="none"):
        self.scope, self.kind, self.size, self.rate, self.act = scope, kind, size, rate, act


TEXTENC_LAYERS = [                                  # networks.py:14-71
    _L("embed_1", "E"),                             # :24-27
    _L("C_2", "C", act="relu"),                     # :28-35
    _L("C_3", "C"),                                 # :36-42
    _L("HC_4", "HC", 3, 1), _L("HC_5", "HC", 3, 3), _L("HC_6", "HC", 3, 9), _L("HC_7", "HC", 3, 27),       # :44-52, first round of j = 0..3 (rate 3^j)
    _L("HC_8", "HC", 3, 1), _L("HC_9", "HC", 3, 3), _L("HC_10", "HC", 3, 9), _L("HC_11", "HC", 3, 27),     # second round
    _L("HC_12", "HC", 3, 1), _L("HC_13", "HC", 3, 1),                                                      # :53-59
    _L("HC_14", "HC", 1, 1), _L("HC_15", "HC", 1, 1),                                                      # :61-67
]
AUDIOENC_LAYERS = [                                 # networks.py:73-124, padding CAUSAL
    _L("C_1", "C", act="relu"), _L("C_2", "C", act="relu"), _L("C_3", "C"),                               # :82-105
    _L("HC_4", "HC", 3, 1), _L("HC_5", "HC", 3, 3), _L("HC_6", "HC", 3, 9), _L("HC_7", "HC", 3, 27),       # :106-114
    _L("HC_8", "HC", 3, 1), _L("HC_9", "HC", 3, 3), _L("HC_10", "HC", 3, 9), _L("HC_11", "HC", 3, 27),
    _L("HC_12", "HC", 3, 3), _L("HC_13", "HC", 3, 3),                                                      # :115-122 (rate 3)
]
AUDIODEC_LAYERS = [                                 # networks.py:157-212, padding CAUSAL
    _L("C_1", "C"),                                                                                        # :167-174
    _L("HC_2", "HC", 3, 1), _L("HC_3", "HC", 3, 3), _L("HC_4", "HC", 3, 9), _L("HC_5", "HC", 3, 27),       # :175-182
    _L("HC_6", "HC", 3, 1), _L("HC_7", "HC", 3, 1),                                                        # :184-191
    _L("C_8", "C", act="relu"), _L("C_9", "C", act="relu"), _L("C_10", "C", act="relu"),                 # :192-200
    _L("C_11", "C"),                                                                                       # :202-209 (sigmoid applied outside, :210)
]
SSRN_LAYERS = [                                     # networks.py:214-292, padding SAME
    _L("C_1", "C"),                                                                                        # :226-232
    _L("HC_2", "HC", 3, 1), _L("HC_3", "HC", 3, 3),                                                        # :233-239
    _L("D_4", "D", 3), _L("HC_5", "HC", 3, 1), _L("HC_6", "HC", 3, 3),                                     # :240-252, first round
    _L("D_7", "D", 3), _L("HC_8", "HC", 3, 1), _L("HC_9", "HC", 3, 3),                                     # second round
    _L("C_10", "C"),                                                                                       # :254-260
    _L("HC_11", "HC", 3, 1), _L("HC_12", "HC", 3, 1),                                                      # :261-267
    _L("C_13", "C"),                                                                                       # :269-275
    _L("C_14", "C", act="relu"), _L("C_15", "C", act="relu"),                                              # :277-284
    _L("C_16", "C"),                                                                                       # :285-290 (sigmoid applied outside, :291)
]


# ----------------------------------------------------------------------------- networks.py, one network forward / backward
def _layer_params(W, sc, L):
    if L.kind == "HC":
        return {"kernel": W[sc + "/conv1d/kernel"], "bias": W[sc + "/conv1d/bias"], "g1": W[sc + "/H1/gamma"], "b1": W[sc + "/H1/beta"],
                "g2": W[sc + "/H2/gamma"], "b2": W[sc + "/H2/beta"]}
    if L.kind == "D":
        return {"kernel": W[sc + "/conv2d_transpose/kernel"], "bias": W[sc + "/conv2d_transpose/bias"], "gamma": W[sc + "/normalize/gamma"], "beta": W[sc + "/normalize/beta"]}
    return {"kernel": W[sc + "/conv1d/kernel"], "bias": W[sc + "/conv1d/bias"], "gamma": W[sc + "/normalize/gamma"], "beta": W[sc + "/normalize/beta"]}


_NAMES = {"HC": {"kernel": "/conv1d/kernel", "bias": "/conv1d/bias", "g1": "/H1/gamma", "b1": "/H1/beta", "g2": "/H2/gamma", "b2": "/H2/beta"},
          "C": {"kernel": "/conv1d/kernel", "bias": "/conv1d/bias", "gamma": "/normalize/gamma", "beta": "/normalize/beta"},
          "D": {"kernel": "/conv2d_transpose/kernel", "bias": "/conv2d_transpose/bias", "gamma": "/normalize/gamma", "beta": "/normalize/beta"}}


def network_forward(layers, W, prefix, x, padding, drop=None):
    """One network of networks.py as its layer list (TEXTENC_LAYERS ... SSRN_LAYERS above): returns (output, inputs of every layer).
    x: the first layer's input (character ids when that layer is the embedding).  drop = (rate, seed, step): training=True, i.e.
    dropout behind every block but the embedding (modules.py:139,195,245)."""
    xs = []
    for li, L in enumerate(layers):
        sc = prefix + "/" + L.scope
        xs.append(x)
        act = None if L.act == "none" else L.act
        if L.kind == "E":
            x = O.embed(x, W[sc + "/lookup_table"])
        elif L.kind == "HC":
            x = hc_fwd(x, _layer_params(W, sc, L), L.rate, padding)
        elif L.kind == "D":
            x = d_fwd(x, _layer_params(W, sc, L))
        else:
            x = c_fwd(x, _layer_params(W, sc, L), L.rate, padding, act)
        if drop is not None and L.kind != "E":
            x = dropout(x, layer_key(drop[1], drop[2], prefix, li), drop[0])
    return x, xs


def network_backward(layers, W, prefix, xs, dy, padding, drop=None):
    """Reverse pass over the same layer list: returns (gradient of the first layer's input or None for an embedding,
    {TF variable name: gradient})."""
    grads, g = {}, dy
    for li, L, xin in zip(reversed(range(len(layers))), reversed(layers), reversed(xs)):
        sc = prefix + "/" + L.scope
        if drop is not None and L.kind != "E":
            g = dropout(g, layer_key(drop[1], drop[2], prefix, li), drop[0])
        act = None if L.act == "none" else L.act
        if L.kind == "E":
            grads[sc + "/lookup_table"] = embed_bwd(xin, g, W[sc + "/lookup_table"].shape[0])
            g = None
            continue
        p = _layer_params(W, sc, L)
        r = hc_bwd(xin, p, g, L.rate, padding) if L.kind == "HC" else (d_bwd(xin, p, g) if L.kind == "D" else c_bwd(xin, p, g, L.rate, padding, act))
        for n, suffix in _NAMES[L.kind].items():
            grads[sc + suffix] = r[n]
        g = r["dx"]
    return g, grads


# ----------------------------------------------------------------------------- utils.py / train.py
def guided_attention(max_N, max_T, g=0.2):
    """utils.py:134-140:  W[n, t] = 1 - exp(-(t / max_T - n / max_N)^2 / (2 g^2))."""
    n = np.arange(max_N, dtype=np.float64)[:, None] / float(max_N)
    t = np.arange(max_T, dtype=np.float64)[None, :] / float(max_T)
    # the reference fills a float32 array from Python-float arithmetic (utils.py:136-139) and the graph holds it as a float32 constant (train.py:41): the
    # table every loss sees is the float32 ROUNDING of the formula (found by the reference pin: the float64 formula differs from it by 3e-9 in loss_att)
    return (1.0 - np.exp(-(t - n) ** 2 / (2.0 * g * g))).astype(np.float32).astype(np.float64)


def learning_rate_decay(init_lr, global_step, warmup_steps=4000.0):
    """utils.py:142-145 (Noam):  lr = init_lr * warmup^0.5 * min(step * warmup^-1.5, step^-0.5), step = global_step + 1."""
    step = float(global_step + 1)
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def sigmoid_xent(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits:  max(x, 0) - x z + log(1 + exp(-|x|))."""
    return np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))


def text2mel_losses(Y, Y_logits, mels, alignments, max_N, max_T):
    """train.py:85-100.  Y, Y_logits, mels (B, T, n_mels); alignments (B, N, T) with N <= max_N, T <= max_T.
    Returns (loss_mels, loss_bd1, loss_att) and the gradients of their SUM with respect to Y (through the L1 term only),
    Y_logits (through the divergence term only) and alignments."""
    loss_mels = np.abs(Y - mels).mean()                                        # :87
    loss_bd1 = sigmoid_xent(Y_logits, mels).mean()                             # :90
    B, N, T = alignments.shape
    A = np.full((B, max_N, max_T), -1.0)                                       # :93  pad with -1 up to (max_N, max_T), then crop
    A[:, :min(N, max_N), :min(T, max_T)] = alignments[:, :max_N, :max_T]
    mask = (A != -1.0).astype(np.float64)                                      # :94
    gts = guided_attention(max_N, max_T)                                       # train.py:30  self.gts
    mask_sum = mask.sum()                                                      # :96
    loss_att = (np.abs(A * gts) * mask).sum() / mask_sum                       # :95-97
    dY = np.sign(Y - mels) / Y.size
    dlogits = (O.sigmoid(Y_logits) - mels) / Y_logits.size
    dA = np.zeros_like(alignments)                                             # entries cropped away (:93) get no gradient
    n_, t_ = min(N, max_N), min(T, max_T)
    dA[:, :n_, :t_] = (np.sign(A * gts) * gts * mask / mask_sum)[:, :n_, :t_]
    return (loss_mels, loss_bd1, loss_att), (dY, dlogits, dA)


def ssrn_losses(Z, Z_logits, mags):
    """train.py:102-110.  Returns (loss_mags, loss_bd2) and the gradients of their sum w.r.t. Z and Z_logits."""
    loss_mags = np.abs(Z - mags).mean()
    loss_bd2 = sigmoid_xent(Z_logits, mags).mean()
    return (loss_mags, loss_bd2), (np.sign(Z - mags) / Z.size, (O.sigmoid(Z_logits) - mags) / Z_logits.size)


def adam_step(var, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """train.py:119-131: gradients clipped element-wise to [-1, 1], then tf.train.AdamOptimizer's update
    (lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  var -= lr_t * m / (sqrt(v) + eps)), t = step (1-based)."""
    g = np.clip(grad, -1.0, 1.0)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    return var - lr_t * m / (np.sqrt(v) + eps), m, v


# ----------------------------------------------------------------------------- train.py:26-134, one training step
def train_grads(num, W, global_step, batch, hp, dropout_seed=None):
    """Forward pass, losses and `optimizer.compute_gradients(loss)` (train.py:119, before the clip) of Graph(num) on one batch:
    returns (losses, {TF variable name: d loss / d variable}).  batch: see train_step."""
    grads = {}
    drop = None if dropout_seed is None else (hp.dropout_rate, dropout_seed, global_step)      # training=True (train.py:55-72)
    if num == 1:
        L, mels = batch
        d = hp.d
        S = np.concatenate((np.zeros_like(mels[:, :1]), mels[:, :-1]), 1)                    # train.py:51
        te, ae, ad = TEXTENC_LAYERS, AUDIOENC_LAYERS, AUDIODEC_LAYERS
        KV, xs_te = network_forward(te, W, "Text2Mel/TextEnc", L, "same", drop)
        K, V = KV[..., :d], KV[..., d:]
        Q, xs_ae = network_forward(ae, W, "Text2Mel/AudioEnc", S, "causal", drop)
        R, al, _ = O.Attention(Q, K, V, hp)
        logits, xs_ad = network_forward(ad, W, "Text2Mel/AudioDec", R, "causal", drop)
        Y = O.sigmoid(logits)
        losses, (dY, dlog, dA) = text2mel_losses(Y, logits, mels, al, hp.max_N, hp.max_T)
        dR, g = network_backward(ad, W, "Text2Mel/AudioDec", xs_ad, dlog + dY * Y * (1 - Y), "causal", drop); grads.update(g)
        dQ, dK, dV = attention_bwd(Q, K, V, dR, dA, d)
        _, g = network_backward(ae, W, "Text2Mel/AudioEnc", xs_ae, dQ, "causal", drop); grads.update(g)
        _, g = network_backward(te, W, "Text2Mel/TextEnc", xs_te, np.concatenate((dK, dV), -1), "same", drop); grads.update(g)
    else:
        mels, mags = batch
        layers = SSRN_LAYERS
        logits, xs = network_forward(layers, W, "SSRN", mels, "same", drop)
        Z = O.sigmoid(logits)
        losses, (dZ, dlog) = ssrn_losses(Z, logits, mags)
        _, grads = network_backward(layers, W, "SSRN", xs, dlog + dZ * Z * (1 - Z), "same", drop)
    return losses, grads


def train_step(num, W, m, v, global_step, batch, hp, dropout_seed=None):
    """One `sess.run(g.train_op)` of train.py for Graph(num): forward, losses, gradients of every variable of the network being
    trained, clip + Adam with the Noam learning rate.  W / m / v: {TF variable name: float64 array}, updated in place.
    batch: num == 1: (L ids (B, N), mels (B, T, n_mels));  num == 2: (mels (B, T, n_mels), mags (B, 4T, n_linear)).
    Returns the losses (loss_mels, loss_bd1, loss_att) or (loss_mags, loss_bd2)."""
    losses, grads = train_grads(num, W, global_step, batch, hp, dropout_seed)
    lr = learning_rate_decay(hp.lr, global_step)                                              # train.py:116
    for n, g in grads.items():
        W[n], m[n], v[n] = adam_step(W[n], g, m[n], v[n], global_step + 1, lr)
    return losses
