"""The training-slice oracle (oracle/train_ref.py) pinned by central finite differences of the float64 forward pass.
CPU only.  The forward pass it differentiates is oracle/dctts_ref.py's (modules.py:143-197)."""
import numpy as np
import pytest

from oracle import dctts_ref as O
from oracle import train_ref as TR


def _params(rng, k, C):
    return {"kernel": rng.normal(0, 0.2, (k, C, 2 * C)), "bias": rng.normal(0, 0.1, 2 * C),
            "g1": 1 + rng.normal(0, 0.1, C), "b1": rng.normal(0, 0.1, C), "g2": 1 + rng.normal(0, 0.1, C), "b2": rng.normal(0, 0.1, C)}


def _fd(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; fp = f()
        x[i] = old - eps; fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
    return g


@pytest.mark.parametrize("k,rate,padding", [(3, 1, "causal"), (3, 3, "causal"), (3, 1, "same"), (3, 3, "same"), (1, 1, "same")])
def test_hc_backward_matches_finite_differences(k, rate, padding):
    rng = np.random.default_rng(5)
    B, T, C = 2, 9, 6
    p = _params(rng, k, C)
    x = rng.normal(0, 1, (B, T, C))
    dy = rng.normal(0, 1, (B, T, C))
    g = TR.hc_bwd(x, p, dy, rate, padding)
    loss = lambda: float((TR.hc_fwd(x, p, rate, padding) * dy).sum())
    np.testing.assert_allclose(g["dx"], _fd(loss, x), rtol=1e-5, atol=1e-7)
    for name in ("kernel", "bias", "g1", "b1", "g2", "b2"):
        np.testing.assert_allclose(g[name], _fd(loss, p[name]), rtol=1e-5, atol=1e-7, err_msg=name)


@pytest.mark.parametrize("k,rate,padding,act", [(1, 1, "same", "relu"), (1, 1, "same", None), (1, 1, "causal", "sigmoid"), (3, 2, "causal", "relu")])
def test_conv1d_backward_matches_finite_differences(k, rate, padding, act):
    rng = np.random.default_rng(15)
    B, T, Ci, Co = 2, 7, 5, 6
    p = {"kernel": rng.normal(0, 0.4, (k, Ci, Co)), "bias": rng.normal(0, 0.1, Co), "gamma": 1 + rng.normal(0, 0.1, Co), "beta": rng.normal(0, 0.3, Co)}
    x = rng.normal(0, 1, (B, T, Ci)); dy = rng.normal(0, 1, (B, T, Co))
    g = TR.c_bwd(x, p, dy, rate, padding, act)
    loss = lambda: float((TR.c_fwd(x, p, rate, padding, act) * dy).sum())
    np.testing.assert_allclose(g["dx"], _fd(loss, x), rtol=1e-5, atol=1e-7)
    for name in ("kernel", "bias", "gamma", "beta"):
        np.testing.assert_allclose(g[name], _fd(loss, p[name]), rtol=1e-5, atol=1e-7, err_msg=name)


def test_conv1d_transpose_backward_matches_finite_differences():
    rng = np.random.default_rng(16)
    B, T, Ci, Co = 2, 5, 4, 6
    p = {"kernel": rng.normal(0, 0.4, (1, 3, Co, Ci)), "bias": rng.normal(0, 0.1, Co), "gamma": 1 + rng.normal(0, 0.1, Co), "beta": rng.normal(0, 0.3, Co)}
    x = rng.normal(0, 1, (B, T, Ci)); dy = rng.normal(0, 1, (B, 2 * T, Co))
    g = TR.d_bwd(x, p, dy)
    loss = lambda: float((TR.d_fwd(x, p) * dy).sum())
    np.testing.assert_allclose(g["dx"], _fd(loss, x), rtol=1e-5, atol=1e-7)
    for name in ("kernel", "bias", "gamma", "beta"):
        np.testing.assert_allclose(g[name], _fd(loss, p[name]), rtol=1e-5, atol=1e-7, err_msg=name)


def test_attention_and_embed_backward_match_finite_differences():
    from dc_tts_amd.hyperparams import hp
    rng = np.random.default_rng(17)
    B, T, N, d = 2, 5, 4, 6
    h = hp.replace(d=d)
    Q, K, V = rng.normal(0, 1, (B, T, d)), rng.normal(0, 1, (B, N, d)), rng.normal(0, 1, (B, N, d))
    dR, dAl = rng.normal(0, 1, (B, T, 2 * d)), rng.normal(0, 1, (B, N, T))
    def loss():
        R, al, _ = O.Attention(Q, K, V, h)
        return float((R * dR).sum() + (al * dAl).sum())
    dQ, dK, dV = TR.attention_bwd(Q, K, V, dR, dAl, d)
    np.testing.assert_allclose(dQ, _fd(loss, Q), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dK, _fd(loss, K), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dV, _fd(loss, V), rtol=1e-5, atol=1e-7)
    ids = rng.integers(0, 5, (3, 7)); table = rng.normal(0, 1, (5, 4)); dy = rng.normal(0, 1, (3, 7, 4))
    f = lambda: float((O.embed(ids, table) * dy).sum())
    np.testing.assert_allclose(TR.embed_bwd(ids, dy, 5), _fd(f, table), rtol=1e-6, atol=1e-8)


def test_normalize_and_conv_backward():
    rng = np.random.default_rng(6)
    x = rng.normal(0, 1, (2, 5, 7)); gam = 1 + rng.normal(0, 0.1, 7); bet = rng.normal(0, 0.1, 7); dy = rng.normal(0, 1, (2, 5, 7))
    dx, dg, db = TR.normalize_bwd(x, gam, dy)
    f = lambda: float((O.normalize(x, gam, bet) * dy).sum())
    np.testing.assert_allclose(dx, _fd(f, x), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dg, _fd(f, gam), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(db, _fd(f, bet), rtol=1e-5, atol=1e-7)
    W = rng.normal(0, 0.3, (3, 7, 4)); b = rng.normal(0, 0.1, 4); dz = rng.normal(0, 1, (2, 5, 4))
    for rate, padding in ((1, "causal"), (2, "same")):
        dx, dW, dbias = TR.conv_bwd(x, W, dz, rate, padding)
        f = lambda: float((O._conv(x, W, b, rate, padding) * dz).sum())
        np.testing.assert_allclose(dx, _fd(f, x), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(dW, _fd(f, W), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(dbias, _fd(f, b), rtol=1e-5, atol=1e-7)


def test_losses_and_their_gradients():
    rng = np.random.default_rng(7)
    B, T, M, N = 2, 6, 5, 4
    max_N, max_T = 7, 9
    logits = rng.normal(0, 2, (B, T, M)); mels = rng.uniform(0, 1, (B, T, M))
    al = rng.uniform(0.01, 1, (B, N, T)); al /= al.sum(axis=1, keepdims=True)
    def total(lg, a):
        (l1, l2, l3), _ = TR.text2mel_losses(O.sigmoid(lg), lg, mels, a, max_N, max_T)
        return l1, l2, l3
    (l1, l2, l3), (dY, dlog, dA) = TR.text2mel_losses(O.sigmoid(logits), logits, mels, al, max_N, max_T)
    # known answers: guided-attention weights (utils.py:134-140) and the cross-entropy identity
    W = TR.guided_attention(max_N, max_T)
    # (the float32 rounding of the formula: the reference stores the table in a float32 array, utils.py:136; pinned by tests/test_reference_pin.py)
    assert W[0, 0] == 0 and W[3, 5] == np.float64(np.float32(1 - np.exp(-(5 / 9 - 3 / 7) ** 2 / 0.08)))
    assert abs(l3 - (al * W[:N, :T]).sum() / (B * N * T)) < 1e-14          # the mask is 1 exactly where alignments exist
    y = O.sigmoid(logits)
    assert abs(l2 - (-(mels * np.log(y) + (1 - mels) * np.log(1 - y))).mean()) < 1e-12
    # gradients: Y enters the L1 term, the logits the divergence term (Y = sigmoid(logits) chains the first into the second)
    f_l1 = lambda: float(np.abs(O.sigmoid(logits) - mels).mean())
    np.testing.assert_allclose(dY * y * (1 - y), _fd(f_l1, logits), rtol=1e-5, atol=1e-9)
    f_bd = lambda: float(TR.sigmoid_xent(logits, mels).mean())
    np.testing.assert_allclose(dlog, _fd(f_bd, logits), rtol=1e-5, atol=1e-9)
    f_att = lambda: total(logits, al)[2]
    np.testing.assert_allclose(dA, _fd(f_att, al), rtol=1e-5, atol=1e-9)
    (m1, m2), (dZ, dZl) = TR.ssrn_losses(y, logits, mels)
    assert abs(m1 - l1) < 1e-15 and abs(m2 - l2) < 1e-15 and np.array_equal(dZ, dY) and np.array_equal(dZl, dlog)


def test_noam_schedule_and_adam_step():
    # utils.py:142-145: linear warm-up to init_lr at step 4000, then step^-0.5
    assert abs(TR.learning_rate_decay(0.001, 3999) - 0.001) < 1e-15
    assert abs(TR.learning_rate_decay(0.001, 0) - 0.001 * 4000 ** 0.5 * 4000 ** -1.5) < 1e-18
    assert abs(TR.learning_rate_decay(0.001, 15999) - 0.0005) < 1e-15
    var = np.array([1.0, -2.0]); g = np.array([5.0, -0.25])
    v1, m, v = TR.adam_step(var, g, np.zeros(2), np.zeros(2), 1, 0.1)
    # first Adam step moves every coordinate by lr * sign(clipped gradient) (up to eps)
    np.testing.assert_allclose(v1, var - 0.1 * np.sign(g), rtol=0, atol=1e-6)
    np.testing.assert_allclose(m, 0.1 * np.clip(g, -1, 1)); np.testing.assert_allclose(v, 0.001 * np.clip(g, -1, 1) ** 2)


def test_backward_oracle_against_torch_autograd():
    """Independent of the finite-difference checks above: the same blocks written with torch.nn.functional (float64, CPU) and
    differentiated by torch.autograd must give oracle/train_ref.py's hand-written gradients (hc, conv1d, conv1d_transpose, attention)."""
    import torch
    import torch.nn.functional as F
    from dc_tts_amd.hyperparams import hp
    rng = np.random.default_rng(21)
    tt = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)

    def t_conv(x, W, b, rate, padding):                       # x (B, T, Ci), W (k, Ci, Co): tf.layers.conv1d as dctts_ref._conv restates it
        k = W.shape[0]
        pl, pr = TR.conv_pads(k, rate, padding)
        y = F.conv1d(F.pad(x.transpose(1, 2), (pl, pr)), W.permute(2, 1, 0), b, dilation=rate)
        return y.transpose(1, 2)

    def t_ln(x, g, b):
        return F.layer_norm(x, (x.shape[-1],), g, b, eps=TR.LN_EPS)

    # hc (CAUSAL, dilation 3) and conv1d (SAME, relu)
    B, T, C = 2, 11, 6
    p = _params(rng, 3, C); x = rng.normal(0, 1, (B, T, C)); dy = rng.normal(0, 1, (B, T, C))
    tx, tp = tt(x), {n: tt(v) for n, v in p.items()}
    H = t_conv(tx, tp["kernel"], tp["bias"], 3, "causal")
    s = torch.sigmoid(t_ln(H[..., :C], tp["g1"], tp["b1"]))
    y = s * t_ln(H[..., C:], tp["g2"], tp["b2"]) + (1 - s) * tx
    (y * torch.tensor(dy)).sum().backward()
    g = TR.hc_bwd(x, p, dy, 3, "causal")
    np.testing.assert_allclose(g["dx"], tx.grad.numpy(), rtol=1e-9, atol=1e-11)
    for n in p:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=n)
    pc = {"kernel": rng.normal(0, 0.4, (1, C, 5)), "bias": rng.normal(0, 0.1, 5), "gamma": 1 + rng.normal(0, 0.1, 5), "beta": rng.normal(0, 0.3, 5)}
    dyc = rng.normal(0, 1, (B, T, 5))
    tx, tp = tt(x), {n: tt(v) for n, v in pc.items()}
    (torch.relu(t_ln(t_conv(tx, tp["kernel"], tp["bias"], 1, "same"), tp["gamma"], tp["beta"])) * torch.tensor(dyc)).sum().backward()
    g = TR.c_bwd(x, pc, dyc, 1, "same", "relu")
    np.testing.assert_allclose(g["dx"], tx.grad.numpy(), rtol=1e-9, atol=1e-11)
    for n in pc:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=n)
    # conv1d_transpose: tf.layers.conv2d_transpose(kernel (1, 3, Cout, Cin), strides (1, 2), 'same') == conv_transpose1d(stride 2,
    # padding 0) cropped to 2T (out[2t] = x[t] W0 + x[t-1] W2, out[2t+1] = x[t] W1: SURVEY B.5)
    pd = {"kernel": rng.normal(0, 0.4, (1, 3, 5, C)), "bias": rng.normal(0, 0.1, 5), "gamma": 1 + rng.normal(0, 0.1, 5), "beta": rng.normal(0, 0.3, 5)}
    dyd = rng.normal(0, 1, (B, 2 * T, 5))
    tx, tp = tt(x), {n: tt(v) for n, v in pd.items()}
    yd = F.conv_transpose1d(tx.transpose(1, 2), tp["kernel"][0].permute(2, 1, 0), tp["bias"], stride=2)[..., :2 * T].transpose(1, 2)
    np.testing.assert_allclose(t_ln(yd, tp["gamma"], tp["beta"]).detach().numpy(), TR.d_fwd(x, pd), rtol=1e-10, atol=1e-12)
    (t_ln(yd, tp["gamma"], tp["beta"]) * torch.tensor(dyd)).sum().backward()
    g = TR.d_bwd(x, pd, dyd)
    np.testing.assert_allclose(g["dx"], tx.grad.numpy(), rtol=1e-9, atol=1e-11)
    for n in pd:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=n)
    # attention (training form) with a gradient on both outputs
    d, N = 6, 5
    Q, K, V = rng.normal(0, 1, (B, T, d)), rng.normal(0, 1, (B, N, d)), rng.normal(0, 1, (B, N, d))
    dR, dAl = rng.normal(0, 1, (B, T, 2 * d)), rng.normal(0, 1, (B, N, T))
    tq, tk, tv = tt(Q), tt(K), tt(V)
    A = torch.softmax(tq @ tk.transpose(1, 2) / np.sqrt(d), -1)
    ((torch.cat((A @ tv, tq), -1) * torch.tensor(dR)).sum() + (A.transpose(1, 2) * torch.tensor(dAl)).sum()).backward()
    dQ, dK, dV = TR.attention_bwd(Q, K, V, dR, dAl, d)
    np.testing.assert_allclose(dQ, tq.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dK, tk.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dV, tv.grad.numpy(), rtol=1e-9, atol=1e-11)


def test_dropout_masks_of_neighbouring_layers_are_independent():
    """tf.layers.dropout draws an independent mask per layer.  The keys of consecutive layers differ by 1 (layer_key), so a counter hash of
    key + i would make layer l+1's mask layer l's shifted by one element; with the key hashed first no small shift lines them up."""
    from oracle import train_ref as TR
    n, rate = 1 << 16, 0.05
    for prefix in ("Text2Mel/AudioEnc", "SSRN"):
        masks = [TR.dropout_mask(TR.layer_key(7, 123, prefix, li), n, rate) for li in range(3)]
        for m in masks:
            assert abs((~m).mean() - rate) < 0.01
        for a, b in ((0, 1), (1, 2), (0, 2)):
            for shift in range(-3, 4):
                x = masks[a][max(0, shift): n + min(0, shift)]
                y = masks[b][max(0, -shift): n - max(0, shift)]
                agree = float((x == y).mean())
                # independent Bernoulli(0.95) masks agree on 0.95^2 + 0.05^2 = 0.905 of the elements
                assert abs(agree - 0.905) < 0.01, (prefix, a, b, shift, agree)
    # and the masks of the same layer at consecutive steps / seeds differ as well
    m0 = TR.dropout_mask(TR.layer_key(7, 123, "SSRN", 0), n, rate)
    m1 = TR.dropout_mask(TR.layer_key(7, 124, "SSRN", 0), n, rate)
    assert abs(float((m0 == m1).mean()) - 0.905) < 0.01
