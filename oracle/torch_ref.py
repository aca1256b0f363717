"""torch-CPU restatement of the reference's synthesize-mode graph -- TEST / BASELINE INFRASTRUCTURE, never the product path.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg may import this module (the product path is HIP: dc_tts_amd/csrc).
BASELINE.md section 3 asks for the CPU baseline in torch-CPU fp32 on all host cores (TensorFlow is not installable here); this is
oracle/dctts_ref.py's arithmetic written on torch.nn.functional, pinned to that numpy oracle by tests/test_oracle.py
(test_torch_restatement_matches_the_numpy_oracle).  Same citations: modules.py:13-247, networks.py:14-292, train.py:48-77,
synthesize.py:45-57.  Layout (B, T, C) like the reference; conv kernels in TF layout (k, Cin, Cout).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def normalize(x, gamma, beta):
    """modules.py:60-63: tf.contrib.layers.layer_norm over the channel axis, eps 1e-12, biased variance."""
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps=1e-12)


def _conv(x, W, b, rate, padding):
    """tf.layers.conv1d, modules.py:120-135: kernel (k, Cin, Cout), SAME / CAUSAL zero padding (modules.py:121-125), cross-correlation."""
    k = W.shape[0]
    total = (k - 1) * rate
    left = total if padding == "causal" else total // 2
    xt = F.pad(x.transpose(1, 2), (left, total - left))                   # (B, C, T)
    y = F.conv1d(xt, W.permute(2, 1, 0).contiguous(), b, dilation=rate)
    return y.transpose(1, 2)


def conv1d(x, P, sc, rate=1, padding="same", act=None):
    y = normalize(_conv(x, P[sc + "/conv1d/kernel"], P[sc + "/conv1d/bias"], rate, padding), P[sc + "/normalize/gamma"], P[sc + "/normalize/beta"])
    return torch.relu(y) if act == "relu" else y                          # modules.py:136-138


def hc(x, P, sc, rate=1, padding="same"):
    """modules.py:143-197: conv to 2C, split, two layer-norms, sigmoid gate, highway mix with the layer's input."""
    t = _conv(x, P[sc + "/conv1d/kernel"], P[sc + "/conv1d/bias"], rate, padding)
    C = x.shape[-1]
    H1 = torch.sigmoid(normalize(t[..., :C], P[sc + "/H1/gamma"], P[sc + "/H1/beta"]))
    H2 = normalize(t[..., C:], P[sc + "/H2/gamma"], P[sc + "/H2/beta"])
    return H1 * H2 + (1.0 - H1) * x


def conv1d_transpose(x, P, sc):
    """modules.py:199-247: conv2d_transpose, kernel (1, 3, Cout, Cin), stride 2, padding 'same' -> exactly 2T rows
    (out[2t] = x[t] W0 + x[t-1] W2, out[2t+1] = x[t] W1), then layer-norm."""
    W = P[sc + "/conv2d_transpose/kernel"][0]                             # (3, Cout, Cin)
    y = F.conv_transpose1d(x.transpose(1, 2), W.permute(2, 1, 0).contiguous(), P[sc + "/conv2d_transpose/bias"], stride=2)[..., :-1]
    return normalize(y.transpose(1, 2), P[sc + "/normalize/gamma"], P[sc + "/normalize/beta"])


def TextEnc(L, P, hp):                                                    # networks.py:14-71
    s = "Text2Mel/TextEnc/"
    tab = P[s + "embed_1/lookup_table"].clone(); tab[0] = 0               # modules.py:36-38
    x = tab[torch.from_numpy(np.asarray(L, np.int64))]
    x = conv1d(x, P, s + "C_2", act="relu"); x = conv1d(x, P, s + "C_3")
    i = 4
    for _ in range(2):
        for j in range(4):
            x = hc(x, P, s + f"HC_{i}", 3 ** j); i += 1
    for _ in range(2):
        x = hc(x, P, s + f"HC_{i}", 1); i += 1
    for _ in range(2):
        x = hc(x, P, s + f"HC_{i}", 1); i += 1                            # k = 1
    return x[..., :hp.d], x[..., hp.d:]


def AudioEnc(S, P, hp):                                                   # networks.py:73-124
    s = "Text2Mel/AudioEnc/"
    x = conv1d(S, P, s + "C_1", padding="causal", act="relu"); x = conv1d(x, P, s + "C_2", padding="causal", act="relu"); x = conv1d(x, P, s + "C_3", padding="causal")
    i = 4
    for _ in range(2):
        for j in range(4):
            x = hc(x, P, s + f"HC_{i}", 3 ** j, "causal"); i += 1
    for _ in range(2):
        x = hc(x, P, s + f"HC_{i}", 3, "causal"); i += 1
    return x


def Attention(Q, K, V, hp, prev_max):                                     # networks.py:126-155, monotonic (synthesis) branch
    A = torch.matmul(Q, K.transpose(1, 2)) * (1.0 / np.sqrt(hp.d))
    N = K.shape[1]
    n = torch.arange(N)[None, :]
    p = torch.from_numpy(np.asarray(prev_max, np.int64))[:, None]
    masked = (n < p) | (n >= p + hp.attention_win_size)
    A = torch.where(masked[:, None, :], torch.full_like(A, float(-2 ** 32 + 1)), A)
    A = torch.softmax(A, -1)
    return torch.cat((torch.matmul(A, V), Q), -1), A, torch.argmax(A, -1)


def AudioDec(R, P, hp):                                                   # networks.py:157-212
    s = "Text2Mel/AudioDec/"
    x = conv1d(R, P, s + "C_1", padding="causal")
    i = 2
    for j in range(4):
        x = hc(x, P, s + f"HC_{i}", 3 ** j, "causal"); i += 1
    for _ in range(2):
        x = hc(x, P, s + f"HC_{i}", 1, "causal"); i += 1
    for _ in range(3):
        x = conv1d(x, P, s + f"C_{i}", padding="causal", act="relu"); i += 1
    logits = conv1d(x, P, s + f"C_{i}", padding="causal")
    return logits, torch.sigmoid(logits)


def SSRN(Y, P, hp):                                                       # networks.py:214-292
    s = "SSRN/"
    x = conv1d(Y, P, s + "C_1")
    i = 2
    for j in range(2):
        x = hc(x, P, s + f"HC_{i}", 3 ** j); i += 1
    for _ in range(2):
        x = conv1d_transpose(x, P, s + f"D_{i}"); i += 1
        for j in range(2):
            x = hc(x, P, s + f"HC_{i}", 3 ** j); i += 1
    x = conv1d(x, P, s + f"C_{i}"); i += 1
    for _ in range(2):
        x = hc(x, P, s + f"HC_{i}", 1); i += 1
    x = conv1d(x, P, s + f"C_{i}"); i += 1
    for _ in range(2):
        x = conv1d(x, P, s + f"C_{i}", act="relu"); i += 1
    logits = conv1d(x, P, s + f"C_{i}")
    return logits, torch.sigmoid(logits)


def params(W):
    return {k: _t(v) for k, v in W.items()}


@torch.no_grad()
def synthesize(L, W, hp, steps=None, run_ssrn=True):
    """synthesize.py:45-57 as the reference runs it: the FULL Text2Mel graph (TextEnc included) at every step, row j kept,
    prev_max fed back; then one SSRN pass.  ``steps`` < max_T stops the loop early (bounded samples for the CPU baseline)."""
    P = params(W)
    B = L.shape[0]
    Y = torch.zeros(B, hp.max_T, hp.n_mels)
    prev = np.zeros((B,), np.int64)
    traj = np.zeros((B, hp.max_T), np.int64)
    for j in range(hp.max_T if steps is None else steps):
        S = torch.cat((torch.zeros_like(Y[:, :1]), Y[:, :-1]), 1)        # train.py:51
        K, V = TextEnc(L, P, hp)
        Q = AudioEnc(S, P, hp)
        R, _, mx = Attention(Q, K, V, hp, prev)
        _, Yj = AudioDec(R, P, hp)
        Y[:, j] = Yj[:, j]
        prev = mx[:, j].numpy()
        traj[:, j] = prev
    Z = SSRN(Y, P, hp)[1] if run_ssrn else None
    return Y.numpy(), (Z.numpy() if Z is not None else None), traj
