"""CPU ORACLE for the DC-TTS synthesis path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``dc_tts_amd``) never does.

PINNING (round 4): the reference (Kyubyong/dc_tts) ships no tests, golden vectors
or fixtures and its arithmetic lives in TensorFlow 1.x (``README.md:7``, not
installable here: no wheel, no network).  What CAN be pinned here is: the
reference's own Python is executed UNMODIFIED on a numpy stand-in for the ~45
TensorFlow symbols it touches (``oracle/tf_shim.py`` + ``oracle/run_reference.py``)
and this file equals that run to 1e-14 in float64 on every tensor of
``train.py:48-80`` (``tests/test_reference_pin.py``); the committed goldens are
generated from the reference run.  So the graph wiring, variable names, shapes,
the loop of ``synthesize.py`` and the text front-end are pinned to the reference
source; the NUMERICS OF TENSORFLOW'S OWN KERNELS (layer_norm's epsilon placement,
SAME / transposed-conv padding, first-index arg-max) remain a restatement from
TF's documented behaviour, shared by this file and the shim
-- "parity partially pinned".  Known-answer tests in ``tests/test_oracle.py`` and
an independent cross-check against torch.nn.functional defend those.

Every function cites the reference lines it restates.  Tensors are channel-last
``(B, time, C)`` exactly like the reference.  ``dtype`` may be float32 (the
reference's arithmetic) or float64 (to separate kernel bugs from fp32
re-association).
"""
import numpy as np

NEG = float(-2 ** 32 + 1)          # networks.py:146  ones_like(A) * (-2 ** 32 + 1)
LN_EPS = 1e-12                      # tf.contrib.layers.layer_norm -> batch_normalization(variance_epsilon=1e-12)


# ----------------------------------------------------------------------------- modules.py
def embed(ids, table):
    """modules.py:13-42  row 0 of the table is replaced by zeros at lookup time (:36-38)."""
    t = table.copy()
    t[0, :] = 0
    return t[ids]


def normalize(x, gamma, beta):
    """modules.py:45-64 -> tf.contrib.layers.layer_norm(begin_norm_axis=-1):
    per (b,t) row, biased two-pass variance (tf.nn.moments), eps 1e-12."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + x.dtype.type(LN_EPS)) * gamma + beta


def _conv(x, W, b, rate, padding):
    """tf.layers.conv1d as called at modules.py:134,187.  W (k, Cin, Cout), cross-correlation.
    CAUSAL = (k-1)*rate zeros on the left then VALID (modules.py:121-125,173-177);
    SAME   = total (k-1)*rate, left = total // 2 (TF SAME rule, stride 1)."""
    k = W.shape[0]
    total = (k - 1) * rate
    if padding.lower() == "causal":
        pl, pr = total, 0
    elif padding.lower() == "same":
        pl = total // 2
        pr = total - pl
    else:
        raise ValueError(padding)
    T = x.shape[1]
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    y = None
    for j in range(k):
        term = xp[:, j * rate: j * rate + T, :] @ W[j]
        y = term if y is None else y + term
    return y + b


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def relu(x):
    return np.maximum(x, 0)


def conv1d(x, P, scope, rate=1, padding="SAME", act=None):
    """modules.py:91-141: conv + bias -> LN (scope 'normalize') -> optional act -> dropout(identity)."""
    y = _conv(x, P[scope + "/conv1d/kernel"], P[scope + "/conv1d/bias"], rate, padding)
    y = normalize(y, P[scope + "/normalize/gamma"], P[scope + "/normalize/beta"])
    if act is not None:
        y = act(y)
    return y


def hc(x, P, scope, rate=1, padding="SAME"):
    """modules.py:143-197: conv to 2C -> split -> LN(H1), LN(H2) -> sigmoid(H1) ->
    H1*H2 + (1-H1)*inputs (activation_fn is None at every call site)."""
    y = _conv(x, P[scope + "/conv1d/kernel"], P[scope + "/conv1d/bias"], rate, padding)
    C = y.shape[-1] // 2
    H1, H2 = y[..., :C], y[..., C:]
    H1 = normalize(H1, P[scope + "/H1/gamma"], P[scope + "/H1/beta"])
    H2 = normalize(H2, P[scope + "/H2/gamma"], P[scope + "/H2/beta"])
    H1 = sigmoid(H1)
    return H1 * H2 + (1.0 - H1) * x


def conv1d_transpose(x, P, scope):
    """modules.py:199-247: tf.layers.conv2d_transpose, kernel (1,3,Cout,Cin), strides (1,2),
    padding 'same' -> length exactly 2T.  Gradient of a SAME stride-2 k=3 conv (pad left 0,
    right 1):  out[2t] = b + x[t] W0 + x[t-1] W2 ;  out[2t+1] = b + x[t] W1   (SURVEY B.5)."""
    W = P[scope + "/conv2d_transpose/kernel"][0]          # (3, Cout, Cin)
    b = P[scope + "/conv2d_transpose/bias"]
    B, T, _ = x.shape
    Co = W.shape[1]
    y = np.zeros((B, 2 * T, Co), x.dtype)
    xm1 = np.pad(x, ((0, 0), (1, 0), (0, 0)))[:, :T, :]
    y[:, 0::2, :] = x @ W[0].T + xm1 @ W[2].T
    y[:, 1::2, :] = x @ W[1].T
    y = y + b
    return normalize(y, P[scope + "/normalize/gamma"], P[scope + "/normalize/beta"])


# ----------------------------------------------------------------------------- networks.py
class _Scoped(dict):
    """View of the weight dict under one variable scope (tf.variable_scope, train.py:49-76)."""

    def __init__(self, W, prefix, dtype):
        super().__init__()
        self.W, self.prefix, self.dtype = W, prefix, dtype

    def __getitem__(self, k):
        return np.asarray(self.W[self.prefix + "/" + k], dtype=self.dtype)


def TextEnc(L, W, hp, dtype=np.float32):
    """networks.py:14-71."""
    P = _Scoped(W, "Text2Mel/TextEnc", dtype)
    i = 1
    x = embed(L, P["embed_1/lookup_table"]); i += 1
    x = conv1d(x, P, f"C_{i}", act=relu); i += 1
    x = conv1d(x, P, f"C_{i}"); i += 1
    for _ in range(2):
        for j in range(4):
            x = hc(x, P, f"HC_{i}", rate=3 ** j); i += 1
    for _ in range(2):
        x = hc(x, P, f"HC_{i}", rate=1); i += 1
    for _ in range(2):
        x = hc(x, P, f"HC_{i}", rate=1); i += 1            # size=1 kernels (shape in the weights)
    d = x.shape[-1] // 2
    return x[..., :d], x[..., d:]


def AudioEnc(S, W, hp, dtype=np.float32):
    """networks.py:73-124 (all CAUSAL)."""
    P = _Scoped(W, "Text2Mel/AudioEnc", dtype)
    i = 1
    x = conv1d(S.astype(dtype), P, f"C_{i}", padding="CAUSAL", act=relu); i += 1
    x = conv1d(x, P, f"C_{i}", padding="CAUSAL", act=relu); i += 1
    x = conv1d(x, P, f"C_{i}", padding="CAUSAL"); i += 1
    for _ in range(2):
        for j in range(4):
            x = hc(x, P, f"HC_{i}", rate=3 ** j, padding="CAUSAL"); i += 1
    for _ in range(2):
        x = hc(x, P, f"HC_{i}", rate=3, padding="CAUSAL"); i += 1
    return x


def Attention(Q, K, V, hp, mononotic_attention=False, prev_max_attentions=None):
    """networks.py:126-155.  Returns (R, alignments (B,N,T), max_attentions (B,T) int64)."""
    dt = Q.dtype
    A = (Q @ K.transpose(0, 2, 1)) * dt.type(1.0 / np.sqrt(dt.type(hp.d)))          # :140
    if mononotic_attention:
        n = np.arange(hp.max_N)[None, :]
        p = np.asarray(prev_max_attentions)[:, None]
        key_masks = n < p                                                            # :142 sequence_mask(p, max_N)
        rev = (n < (hp.max_N - hp.attention_win_size - p))[:, ::-1]                  # :143
        masks = np.logical_or(key_masks, rev)                                        # :144
        masks = np.tile(masks[:, None, :], (1, hp.max_T, 1))                         # :145
        A = np.where(masks == False, A, dt.type(NEG))                                # :146-147  # noqa: E712
    A = A - A.max(axis=-1, keepdims=True)                                            # :148 softmax
    A = np.exp(A)
    A = A / A.sum(axis=-1, keepdims=True)
    max_attentions = np.argmax(A, -1).astype(np.int64)                               # :149 (first index on ties)
    R = A @ V                                                                        # :150
    R = np.concatenate((R, Q), -1)                                                   # :151
    alignments = A.transpose(0, 2, 1)                                                # :153
    return R, alignments, max_attentions


def AudioDec(R, W, hp, dtype=np.float32):
    """networks.py:157-212 (all CAUSAL).  Returns (logits, sigmoid(logits))."""
    P = _Scoped(W, "Text2Mel/AudioDec", dtype)
    i = 1
    x = conv1d(R.astype(dtype), P, f"C_{i}", padding="CAUSAL"); i += 1
    for j in range(4):
        x = hc(x, P, f"HC_{i}", rate=3 ** j, padding="CAUSAL"); i += 1
    for _ in range(2):
        x = hc(x, P, f"HC_{i}", rate=1, padding="CAUSAL"); i += 1
    for _ in range(3):
        x = conv1d(x, P, f"C_{i}", padding="CAUSAL", act=relu); i += 1
    logits = conv1d(x, P, f"C_{i}", padding="CAUSAL"); i += 1
    return logits, sigmoid(logits)


def SSRN(Y, W, hp, dtype=np.float32):
    """networks.py:214-292 (all SAME).  Returns (logits, sigmoid(logits)), (B, 4T, 1+n_fft/2)."""
    P = _Scoped(W, "SSRN", dtype)
    i = 1
    x = conv1d(Y.astype(dtype), P, f"C_{i}"); i += 1
    for j in range(2):
        x = hc(x, P, f"HC_{i}", rate=3 ** j); i += 1
    for _ in range(2):
        x = conv1d_transpose(x, P, f"D_{i}"); i += 1
        for j in range(2):
            x = hc(x, P, f"HC_{i}", rate=3 ** j); i += 1
    x = conv1d(x, P, f"C_{i}"); i += 1
    for _ in range(2):
        x = hc(x, P, f"HC_{i}", rate=1); i += 1
    x = conv1d(x, P, f"C_{i}"); i += 1
    for _ in range(2):
        x = conv1d(x, P, f"C_{i}", act=relu); i += 1
    logits = conv1d(x, P, f"C_{i}")
    return logits, sigmoid(logits)


# ----------------------------------------------------------------------------- train.py / synthesize.py
def text2mel_graph(L, mels, prev_max_attentions, W, hp, dtype=np.float32, KV=None):
    """One evaluation of the synthesize-mode Text2Mel graph, train.py:48-68:
    S = concat(zeros, mels[:, :-1]) (:51) -> TextEnc -> AudioEnc -> Attention(monotonic) -> AudioDec.
    ``KV`` may carry a previous TextEnc result: TextEnc is a pure function of L, so re-running it
    (as the reference does at every step) returns the identical tensors."""
    S = np.concatenate((np.zeros_like(mels[:, :1, :]), mels[:, :-1, :]), 1)
    K, V = KV if KV is not None else TextEnc(L, W, hp, dtype)
    Q = AudioEnc(S, W, hp, dtype)
    R, alignments, max_att = Attention(Q, K, V, hp, True, prev_max_attentions)
    logits, Y = AudioDec(R, W, hp, dtype)
    return dict(K=K, V=V, Q=Q, R=R, alignments=alignments, max_attentions=max_att, Y_logits=logits, Y=Y)


def synthesize(L, W, hp, dtype=np.float32, recompute_textenc=False, run_ssrn=True, trace=None, prev0=None):
    """synthesize.py:45-57: 210 x full Text2Mel graph, keep row j, feed prev_max; then one SSRN pass.
    ``prev0`` (tests only) replaces the zeros the reference starts prev_max_attentions from (:46), so that the end-of-text
    window regimes can be reached at max_N = 180 with random weights.

    Returns (Y (B,max_T,n_mels), Z (B,4*max_T,1025) or None, max_att_trajectory (B,max_T) int64)."""
    B = L.shape[0]
    Y = np.zeros((B, hp.max_T, hp.n_mels), dtype)                                   # :45
    prev = np.zeros((B,), np.int32) if prev0 is None else np.asarray(prev0, np.int32).copy()   # :46
    traj = np.zeros((B, hp.max_T), np.int64)
    KV = None
    for j in range(hp.max_T):                                                       # :47
        g = text2mel_graph(L, Y, prev, W, hp, dtype, None if recompute_textenc else KV)
        KV = (g["K"], g["V"])
        Y[:, j, :] = g["Y"][:, j, :]                                                # :53
        prev = g["max_attentions"][:, j].astype(np.int32)                           # :54
        traj[:, j] = g["max_attentions"][:, j]
        if trace is not None:
            trace(j, g)
    Z = SSRN(Y, W, hp, dtype)[1] if run_ssrn else None                              # :57
    return Y, Z, traj


# ----------------------------------------------------------------------------- data_load.py (synthesize branch)
def text_normalize(text, vocab):
    """data_load.py:24-31."""
    import re
    import unicodedata
    text = "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")
    text = text.lower()
    text = re.sub("[^{}]".format(vocab), " ", text)
    text = re.sub("[ ]+", " ", text)
    return text


def load_sentences(lines, hp):
    """data_load.py:79-86 given the lines of the test file AFTER the header line."""
    char2idx = {c: i for i, c in enumerate(hp.vocab)}
    sents = [text_normalize(line.split(" ", 1)[-1], hp.vocab).strip() + "E" for line in lines]
    texts = np.zeros((len(sents), hp.max_N), np.int32)
    for i, s in enumerate(sents):
        texts[i, :len(s)] = [char2idx[c] for c in s]
    return texts
