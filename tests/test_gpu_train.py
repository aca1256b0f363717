"""GPU parity of the first training slice (include/dctts_train.h) against oracle/train_ref.py (float64 numpy, itself pinned by
finite differences in tests/test_train_oracle.py).  Tolerances are relative to the largest magnitude of each gradient: the HIP
path is fp32 (MFMA contractions over up to 7 000 rows), the oracle float64."""
import os

import numpy as np
import pytest
import torch

from oracle import train_ref as TR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dc_tts_amd.train import TrainOps
    o = TrainOps()
    yield o
    o.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("B,T,C,k,rate,padding", [
    (2, 37, 256, 3, 1, "causal"),      # AudioEnc / AudioDec HC, dilation 1, ragged T
    (3, 50, 256, 3, 9, "causal"),      # dilation 9
    (2, 64, 256, 3, 27, "causal"),     # dilation 27 > T / 3: most taps read padding
    (2, 41, 512, 3, 3, "same"),        # TextEnc / SSRN HC
    (2, 33, 512, 1, 1, "same"),        # TextEnc HC_15/16 (k = 1)
    (1, 24, 1024, 3, 1, "same"),       # SSRN HC_11/12
])
def test_hc_backward_vs_oracle(ops, B, T, C, k, rate, padding):
    rng = np.random.default_rng(100 + C + rate)
    p = {"kernel": rng.normal(0, (k * C) ** -0.5, (k, C, 2 * C)), "bias": rng.normal(0, 0.1, 2 * C),
         "g1": 1 + rng.normal(0, 0.1, C), "b1": rng.normal(0, 0.1, C), "g2": 1 + rng.normal(0, 0.1, C), "b2": rng.normal(0, 0.1, C)}
    p = {n: v.astype(np.float32).astype(np.float64) for n, v in p.items()}
    x = rng.normal(0, 1, (B, T, C)).astype(np.float32).astype(np.float64)
    dy = rng.normal(0, 1, (B, T, C)).astype(np.float32).astype(np.float64)
    ref = TR.hc_bwd(x, p, dy, rate, padding)
    got = ops.hc_backward(dev(x), dev(dy), {n: dev(v) for n, v in p.items()}, rate=rate, padding=padding)
    torch.cuda.synchronize()
    for name in ("dx", "kernel", "bias", "g1", "b1", "g2", "b2"):
        e = rel(got[name].cpu().numpy().astype(np.float64), ref[name])
        assert e < 2e-5, f"{name}: relative error {e}"


@pytest.mark.parametrize("B,T,Cin,Cout,k,rate,padding,act", [
    (2, 37, 80, 256, 1, 1, "causal", "relu"),     # AudioEnc C_1 (mel -> d)
    (3, 29, 256, 256, 1, 1, "causal", None),      # AudioEnc C_3
    (2, 45, 128, 512, 1, 1, "same", "relu"),      # TextEnc C_2 (e -> 2d)
    (2, 31, 512, 1024, 1, 1, "same", None),       # SSRN C_10 (c -> 2c)
    (2, 33, 256, 256, 3, 2, "causal", "sigmoid"), # not a layer of the model: the generic k = 3 / sigmoid paths
    (2, 21, 256, 80, 1, 1, "causal", None),       # AudioDec C_11 (d -> n_mels): a width that is not a multiple of 256
    (1, 12, 1024, 1025, 1, 1, "same", None),      # SSRN C_13 (2c -> 1 + n_fft / 2): not a multiple of 4 either
    (1, 10, 1025, 1025, 1, 1, "same", "relu"),    # SSRN C_14 / C_15
])
def test_conv1d_backward_vs_oracle(ops, B, T, Cin, Cout, k, rate, padding, act):
    rng = np.random.default_rng(200 + Cin + Cout)
    p = {"kernel": rng.normal(0, (k * Cin) ** -0.5, (k, Cin, Cout)), "bias": rng.normal(0, 0.1, Cout), "gamma": 1 + rng.normal(0, 0.1, Cout), "beta": rng.normal(0, 0.3, Cout)}
    p = {n: v.astype(np.float32).astype(np.float64) for n, v in p.items()}
    x = rng.normal(0, 1, (B, T, Cin)).astype(np.float32).astype(np.float64)
    dy = rng.normal(0, 1, (B, T, Cout)).astype(np.float32).astype(np.float64)
    ref = TR.c_bwd(x, p, dy, rate, padding, act)
    got = ops.conv1d_backward(dev(x), dev(dy), {n: dev(v) for n, v in p.items()}, rate=rate, padding=padding, act=act)
    torch.cuda.synchronize()
    for name in ("dx", "kernel", "bias", "gamma", "beta"):
        e = rel(got[name].cpu().numpy().astype(np.float64), ref[name])
        assert e < 2e-5, f"{name}: relative error {e}"


@pytest.mark.parametrize("B,T,C", [(2, 23, 512), (3, 16, 256)])
def test_conv1d_transpose_backward_vs_oracle(ops, B, T, C):
    rng = np.random.default_rng(300 + C)
    p = {"kernel": rng.normal(0, (3 * C) ** -0.5, (1, 3, C, C)), "bias": rng.normal(0, 0.1, C), "gamma": 1 + rng.normal(0, 0.1, C), "beta": rng.normal(0, 0.3, C)}
    p = {n: v.astype(np.float32).astype(np.float64) for n, v in p.items()}
    x = rng.normal(0, 1, (B, T, C)).astype(np.float32).astype(np.float64)
    dy = rng.normal(0, 1, (B, 2 * T, C)).astype(np.float32).astype(np.float64)
    ref = TR.d_bwd(x, p, dy)
    got = ops.conv1d_transpose_backward(dev(x), dev(dy), {n: dev(v) for n, v in p.items()})
    torch.cuda.synchronize()
    for name in ("dx", "kernel", "bias", "gamma", "beta"):
        e = rel(got[name].cpu().numpy().astype(np.float64), ref[name])
        assert e < 2e-5, f"{name}: relative error {e}"


@pytest.mark.parametrize("N", [36, 37, 38, 39])
def test_attention_and_embed_backward_vs_oracle(ops, N):
    """Any text length N: the attention matrix keeps a leading dimension rounded up to 4 inside the library, the softmax and every product run over the
    true N (the reference pads a batch to its longest text, whatever that is: data_load.py:152-160)."""
    rng = np.random.default_rng(41)
    B, T, d = 3, 50, 256
    f = lambda *s: rng.normal(0, 1, s).astype(np.float32).astype(np.float64)
    Q, K, V, dR, dAl = f(B, T, d), f(B, N, d), f(B, N, d), f(B, T, 2 * d), f(B, N, T)
    rQ, rK, rV = TR.attention_bwd(Q, K, V, dR, dAl, d)
    gQ, gK, gV = ops.attention_backward(dev(Q), dev(K), dev(V), dev(dR), dev(dAl))
    torch.cuda.synchronize()
    assert rel(gQ.cpu().numpy(), rQ) < 2e-5 and rel(gK.cpu().numpy(), rK) < 2e-5 and rel(gV.cpu().numpy(), rV) < 2e-5
    from oracle import dctts_ref as O
    from dc_tts_amd.hyperparams import hp as hp_
    Rr, alr, _ = O.Attention(Q, K, V, hp_)
    Rg, alg = ops.attention_forward(dev(Q), dev(K), dev(V))
    torch.cuda.synchronize()
    assert float(np.abs(Rg.cpu().numpy() - Rr).max()) < 1e-4 and float(np.abs(alg.cpu().numpy() - alr).max()) < 1e-5 and tuple(alg.shape) == (B, N, T)
    ids = rng.integers(0, 32, (4, 60)).astype(np.int32); dy = f(4, 60, 128)
    gT = ops.embed_backward(torch.from_numpy(ids).cuda(), dev(dy), 32)
    torch.cuda.synchronize()
    assert rel(gT.cpu().numpy(), TR.embed_bwd(ids, dy, 32)) < 1e-5 and float(gT[0].abs().max()) == 0.0


def test_audioenc_backward_end_to_end(ops, weights):
    """A whole network's backward pass on the GPU: AudioEnc (networks.py:73-124: 3 conv1d + 10 highway blocks, all CAUSAL), layer by
    layer in reverse with the HIP kernels, against the float64 oracle chain: d(loss)/d(S) and every parameter gradient.  The layer
    inputs come from the oracle's forward pass, so this test is about the backward pass only."""
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.layers import audioenc_layers
    rng = np.random.default_rng(31)
    B, T = 2, 40
    S = rng.uniform(0, 1, (B, T, hp.n_mels))
    dQ = rng.normal(0, 1, (B, T, hp.d))
    pref = "Text2Mel/AudioEnc/"
    layers = audioenc_layers(hp)                                  # Layer(scope, kind, cin, cout, size, rate, act) in network order
    W = {n: np.asarray(v, np.float64) for n, v in weights.items() if n.startswith(pref)}
    def params(L):
        sc = pref + L.scope
        if L.kind == "HC":
            return {"kernel": W[sc + "/conv1d/kernel"], "bias": W[sc + "/conv1d/bias"], "g1": W[sc + "/H1/gamma"], "b1": W[sc + "/H1/beta"],
                    "g2": W[sc + "/H2/gamma"], "b2": W[sc + "/H2/beta"]}
        return {"kernel": W[sc + "/conv1d/kernel"], "bias": W[sc + "/conv1d/bias"], "gamma": W[sc + "/normalize/gamma"], "beta": W[sc + "/normalize/beta"]}
    act_of = lambda L: None if L.act == "none" else L.act
    xs, x = [], S
    for L in layers:                                              # forward, keeping every layer's input
        xs.append(x)
        x = TR.hc_fwd(x, params(L), L.rate, "causal") if L.kind == "HC" else TR.c_fwd(x, params(L), L.rate, "causal", act_of(L))
    g_ref, g_gpu = dQ, dev(dQ)
    worst = 0.0
    for L, xin in zip(reversed(layers), reversed(xs)):
        p = params(L)
        if L.kind == "HC":
            r = TR.hc_bwd(xin, p, g_ref, L.rate, "causal")
            o = ops.hc_backward(dev(xin), g_gpu, {n: dev(v) for n, v in p.items()}, rate=L.rate, padding="causal")
        else:
            r = TR.c_bwd(xin, p, g_ref, L.rate, "causal", act_of(L))
            o = ops.conv1d_backward(dev(xin), g_gpu, {n: dev(v) for n, v in p.items()}, rate=L.rate, padding="causal", act=act_of(L))
        for n in p:
            worst = max(worst, rel(o[n].cpu().numpy().astype(np.float64), r[n]))
        g_ref, g_gpu = r["dx"], o["dx"]
    torch.cuda.synchronize()
    e = rel(g_gpu.cpu().numpy().astype(np.float64), g_ref)
    assert e < 1e-4 and worst < 1e-4, (e, worst)                  # 13 layers of fp32 error accumulate on the way down


def _small_weights(h, seed):
    """Seeded weights of the whole model at hyper-parameters h, rounded to fp32 and widened (both sides see the same numbers)."""
    from dc_tts_amd.weights import synthetic_weights
    return {n: np.asarray(v, np.float32).astype(np.float64) for n, v in synthetic_weights(h, seed=seed, perturb=True).items()}


def _compare(grads_gpu, grads_ref, tol):
    worst, worst_name = 0.0, ""
    assert set(grads_gpu) == set(grads_ref)
    for n, r in grads_ref.items():
        e = rel(grads_gpu[n].cpu().numpy().astype(np.float64), r)
        if e > worst:
            worst, worst_name = e, n
    assert worst < tol, (worst_name, worst)
    return worst


def test_ssrn_training_gradients_end_to_end(ops):
    """train.py num == 2: SSRN(mels) -> loss_mags + loss_bd2 (train.py:102-110) -> every SSRN parameter gradient, on the GPU (losses
    + the reverse pass over all 18 layers incl. both transposed convolutions and the 1025-channel output layers) against the float64
    oracle.  Layer inputs come from the oracle's forward pass."""
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.layers import ssrn_layers
    from dc_tts_amd.train import network_backward
    from oracle import dctts_ref as O
    rng = np.random.default_rng(51)
    W = _small_weights(hp, 77)
    B, T = 1, 6
    mels = rng.uniform(0, 1, (B, T, hp.n_mels)).astype(np.float32).astype(np.float64)
    mags = rng.uniform(0, 1, (B, 4 * T, hp.n_linear)).astype(np.float32).astype(np.float64)
    layers = ssrn_layers(hp)
    logits, xs = TR.network_forward(layers, W, "SSRN", mels, "same")
    Z = O.sigmoid(logits)
    (l1, l2), (dZ, dlog) = TR.ssrn_losses(Z, logits, mags)
    _, gref = TR.network_backward(layers, W, "SSRN", xs, dlog + dZ * Z * (1 - Z), "same")
    Wd = {n: dev(v) for n, v in W.items() if n.startswith("SSRN/")}
    losses, gZ, glog = ops.ssrn_losses(dev(Z), dev(logits), dev(mags))
    Zd = dev(Z)
    _, ggpu = network_backward(ops, layers, Wd, "SSRN", [dev(x) for x in xs], glog + gZ * Zd * (1 - Zd), "same")
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses.cpu().numpy(), [l1, l2], rtol=2e-5)
    _compare(ggpu, gref, 2e-4)


def test_text2mel_training_gradients_end_to_end(ops):
    """train.py num == 1: TextEnc, AudioEnc, Attention (training form), AudioDec, loss_mels + loss_bd1 + loss_att (train.py:49-100)
    -> the gradient of every Text2Mel variable (embedding table included), GPU against the float64 oracle."""
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.layers import audiodec_layers, audioenc_layers, textenc_layers
    from dc_tts_amd.train import network_backward
    from oracle import dctts_ref as O
    rng = np.random.default_rng(52)
    W = _small_weights(hp, 78)
    B, N, T, d = 2, 12, 10, hp.d
    ids = rng.integers(1, len(hp.vocab), (B, N)).astype(np.int32); ids[:, -2:] = 0
    mels = rng.uniform(0, 1, (B, T, hp.n_mels)).astype(np.float32).astype(np.float64)
    S = np.concatenate((np.zeros_like(mels[:, :1]), mels[:, :-1]), 1)                       # train.py:51
    te, ae, ad = textenc_layers(hp), audioenc_layers(hp), audiodec_layers(hp)
    KV, xs_te = TR.network_forward(te, W, "Text2Mel/TextEnc", ids, "same")
    K, V = KV[..., :d], KV[..., d:]
    Q, xs_ae = TR.network_forward(ae, W, "Text2Mel/AudioEnc", S, "causal")
    R, al, _ = O.Attention(Q, K, V, hp)
    logits, xs_ad = TR.network_forward(ad, W, "Text2Mel/AudioDec", R, "causal")
    Y = O.sigmoid(logits)
    (l1, l2, l3), (dY, dlog, dA) = TR.text2mel_losses(Y, logits, mels, al, hp.max_N, hp.max_T)
    gref = {}
    dR, g = TR.network_backward(ad, W, "Text2Mel/AudioDec", xs_ad, dlog + dY * Y * (1 - Y), "causal"); gref.update(g)
    dQ, dK, dV = TR.attention_bwd(Q, K, V, dR, dA, d)
    _, g = TR.network_backward(ae, W, "Text2Mel/AudioEnc", xs_ae, dQ, "causal"); gref.update(g)
    _, g = TR.network_backward(te, W, "Text2Mel/TextEnc", xs_te, np.concatenate((dK, dV), -1), "same"); gref.update(g)
    # the same on the GPU
    Wd = {n: dev(v) for n, v in W.items() if n.startswith("Text2Mel/")}
    to = lambda xs: [torch.from_numpy(x).cuda() if x.dtype == np.int32 else dev(x) for x in xs]
    Yd = dev(Y)
    losses, gY, glog, gA = ops.text2mel_losses(Yd, dev(logits), dev(mels), dev(al), hp.max_N, hp.max_T)
    ggpu = {}
    gR, g = network_backward(ops, ad, Wd, "Text2Mel/AudioDec", to(xs_ad), glog + gY * Yd * (1 - Yd), "causal"); ggpu.update(g)
    gQ, gK, gV = ops.attention_backward(dev(Q), dev(K), dev(V), gR, gA)
    _, g = network_backward(ops, ae, Wd, "Text2Mel/AudioEnc", to(xs_ae), gQ, "causal"); ggpu.update(g)
    _, g = network_backward(ops, te, Wd, "Text2Mel/TextEnc", to(xs_te), torch.cat((gK, gV), -1).contiguous(), "same"); ggpu.update(g)
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses.cpu().numpy(), [l1, l2, l3], rtol=2e-5)
    assert len(gref) == len([n for n in W if n.startswith("Text2Mel/")])                   # every Text2Mel variable has a gradient
    _compare(ggpu, gref, 5e-4)


def test_forward_blocks_vs_oracle(ops):
    """The trainer's own forward passes (TF-layout variables) against the oracle's forward pass."""
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.layers import audioenc_layers, ssrn_layers
    from dc_tts_amd.train import network_forward
    from oracle import dctts_ref as O
    rng = np.random.default_rng(61)
    W = _small_weights(hp, 79)
    Wd = {n: dev(v) for n, v in W.items()}
    mels = rng.uniform(0, 1, (2, 9, hp.n_mels)).astype(np.float32).astype(np.float64)
    for layers, prefix, pad in ((audioenc_layers(hp), "Text2Mel/AudioEnc", "causal"), (ssrn_layers(hp), "SSRN", "same")):
        yr, _ = TR.network_forward(layers, W, prefix, mels, pad)
        yg, _ = network_forward(ops, layers, Wd, prefix, dev(mels), pad)
        torch.cuda.synchronize()
        assert maxabs(yg.cpu().numpy(), yr) < 1e-3, prefix
    Q, K, V = (rng.normal(0, 1, s).astype(np.float32).astype(np.float64) for s in ((2, 9, 256), (2, 8, 256), (2, 8, 256)))
    Rr, alr, _ = O.Attention(Q, K, V, hp)
    Rg, alg = ops.attention_forward(dev(Q), dev(K), dev(V))
    torch.cuda.synchronize()
    assert maxabs(Rg.cpu().numpy(), Rr) < 1e-4 and maxabs(alg.cpu().numpy(), alr) < 1e-5


def test_dropout_mask_matches_its_restatement(ops):
    """tf.layers.dropout's random stream cannot be reproduced; what is checked is that the device draws the mask oracle/train_ref.py
    restates (so the oracle can follow a training step with dropout), that about `rate` of the elements are dropped, and the scaling."""
    from dc_tts_amd.train import layer_key
    x = torch.ones(3, 70, 256, device="cuda")
    key = layer_key(5, 4000, "Text2Mel/AudioEnc", 7)
    assert key == TR.layer_key(5, 4000, "Text2Mel/AudioEnc", 7)
    y = ops.dropout(x, key, 0.05).cpu().numpy()
    keep = TR.dropout_mask(key, x.numel(), 0.05).reshape(y.shape)
    assert np.array_equal(y != 0, keep) and abs(float(keep.mean()) - 0.95) < 0.01
    np.testing.assert_allclose(y[keep], 1.0 / 0.95, rtol=1e-6)


def maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max())


def test_train_steps_vs_oracle():
    """TrainGraph.train_op (dc_tts_amd/train.py) = one sess.run(g.train_op) of train.py: two steps of Text2Mel and one of SSRN on a fixed
    tiny batch against the float64 restatement (oracle/train_ref.train_step): the losses of every step and every updated variable."""
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.train import TrainGraph
    rng = np.random.default_rng(71)
    W0 = _small_weights(hp, 80)
    GS0 = 3999
    # ---- Text2Mel, two steps
    B, N, T = 2, 12, 10
    ids = rng.integers(1, len(hp.vocab), (B, N)).astype(np.int32)
    mels = rng.uniform(0, 1, (B, T, hp.n_mels)).astype(np.float32).astype(np.float64)
    Wr = {n: v.copy() for n, v in W0.items() if n.startswith("Text2Mel/")}
    mr = {n: np.zeros_like(v) for n, v in Wr.items()}; vr = {n: np.zeros_like(v) for n, v in Wr.items()}
    g = TrainGraph(1, W0, hp, training=True, seed=5)
    g.global_step = GS0                       # at the peak of the Noam schedule (lr = 1e-3): the updates are far above fp32 resolution
    for step in range(GS0, GS0 + 2):
        lr_ = TR.train_step(1, Wr, mr, vr, step, (ids, mels), hp, dropout_seed=5)
        lg = g.train_op(torch.from_numpy(ids).cuda(), dev(mels))
        torch.cuda.synchronize()
        np.testing.assert_allclose(lg.cpu().numpy(), lr_, rtol=2e-4)
    assert g.global_step == GS0 + 2
    # every variable moved by ~lr per step in the direction the oracle says (Adam's first steps are sign-like): compare the updates
    for n in Wr:
        upd_r = Wr[n] - W0[n]; upd_g = g.W[n].cpu().numpy().astype(np.float64) - W0[n]
        assert np.abs(upd_g - upd_r).max() < 0.05 * np.abs(upd_r).max() + 2e-7, n
    # ---- SSRN, one step
    B, T = 1, 4
    mels = rng.uniform(0, 1, (B, T, hp.n_mels)).astype(np.float32).astype(np.float64)
    mags = rng.uniform(0, 1, (B, 4 * T, hp.n_linear)).astype(np.float32).astype(np.float64)
    Wr = {n: v.copy() for n, v in W0.items() if n.startswith("SSRN/")}
    mr = {n: np.zeros_like(v) for n, v in Wr.items()}; vr = {n: np.zeros_like(v) for n, v in Wr.items()}
    g2 = TrainGraph(2, W0, hp, training=False)              # the dropout-free graph
    g2.global_step = GS0
    lr_ = TR.train_step(2, Wr, mr, vr, GS0, (mels, mags), hp)
    lg = g2.train_op(dev(mels), dev(mags))
    torch.cuda.synchronize()
    np.testing.assert_allclose(lg.cpu().numpy(), lr_, rtol=5e-5)
    for n in Wr:
        upd_r = Wr[n] - W0[n]; upd_g = g2.W[n].cpu().numpy().astype(np.float64) - W0[n]
        assert np.abs(upd_g - upd_r).max() < 0.05 * np.abs(upd_r).max() + 2e-7, n


def test_hc_backward_is_reproducible_and_rejects_bad_shapes(ops):
    rng = np.random.default_rng(3)
    C = 256
    p = {"kernel": dev(rng.normal(0, 0.05, (3, C, 2 * C))), "bias": dev(rng.normal(0, 0.1, 2 * C)), "g1": dev(np.ones(C)), "b1": dev(np.zeros(C)),
         "g2": dev(np.ones(C)), "b2": dev(np.zeros(C))}
    x, dy = dev(rng.normal(0, 1, (4, 100, C))), dev(rng.normal(0, 1, (4, 100, C)))
    a = ops.hc_backward(x, dy, p, rate=3, padding="causal")
    b = ops.hc_backward(x, dy, p, rate=3, padding="causal")
    torch.cuda.synchronize()
    for n in a:
        assert torch.equal(a[n], b[n]), f"{n}: two-stage reductions must be bitwise reproducible"
    from dc_tts_amd.engine import DcttsError
    with pytest.raises(ValueError):
        ops.hc_backward(x, dy[:, :50], p)
    with pytest.raises(DcttsError):
        ops.hc_backward(dev(rng.normal(0, 1, (1, 8, 128))), dev(rng.normal(0, 1, (1, 8, 128))),
                        {"kernel": dev(np.zeros((3, 128, 256))), "bias": dev(np.zeros(256)), "g1": dev(np.ones(128)), "b1": dev(np.zeros(128)),
                         "g2": dev(np.ones(128)), "b2": dev(np.zeros(128))})


def test_losses_vs_oracle(ops):
    rng = np.random.default_rng(8)
    B, T, M, N, max_N, max_T = 3, 50, 80, 40, 180, 210
    logits = rng.normal(0, 2, (B, T, M)).astype(np.float32).astype(np.float64)
    Y = 1 / (1 + np.exp(-logits))
    mels = rng.uniform(0, 1, (B, T, M)).astype(np.float32).astype(np.float64)
    al = rng.uniform(0.0, 1, (B, N, T)); al /= al.sum(axis=1, keepdims=True); al = al.astype(np.float32).astype(np.float64)
    (l1, l2, l3), (dY, dlog, dA) = TR.text2mel_losses(Y, logits, mels, al, max_N, max_T)
    losses, gY, glog, gA = ops.text2mel_losses(dev(Y), dev(logits), dev(mels), dev(al), max_N, max_T)
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses.cpu().numpy(), [l1, l2, l3], rtol=2e-5)
    assert rel(gY.cpu().numpy(), dY) < 1e-6 and rel(glog.cpu().numpy(), dlog) < 2e-5 and rel(gA.cpu().numpy(), dA) < 2e-5
    # alignments larger than (max_N, max_T): the reference crops them (train.py:93)
    (_, _, l3c), (_, _, dAc) = TR.text2mel_losses(Y, logits, mels, al, 30, 45)
    lossesc, _, _, gAc = ops.text2mel_losses(dev(Y), dev(logits), dev(mels), dev(al), 30, 45)
    torch.cuda.synchronize()
    assert abs(float(lossesc[2]) - l3c) < 2e-5 * l3c and rel(gAc.cpu().numpy(), dAc) < 2e-5
    F = 1025
    Zl = rng.normal(0, 2, (2, 40, F)).astype(np.float32).astype(np.float64); Z = 1 / (1 + np.exp(-Zl)); mags = rng.uniform(0, 1, (2, 40, F))
    (m1, m2), (dZ, dZl) = TR.ssrn_losses(Z, Zl, mags.astype(np.float32).astype(np.float64))
    ls, gZ, gZl = ops.ssrn_losses(dev(Z), dev(Zl), dev(mags))
    torch.cuda.synchronize()
    np.testing.assert_allclose(ls.cpu().numpy(), [m1, m2], rtol=2e-5)
    assert rel(gZ.cpu().numpy(), dZ) < 1e-6 and rel(gZl.cpu().numpy(), dZl) < 2e-5


def test_adam_multi_equals_per_variable_updates(ops):
    """One launch over a list of variables (sizes from 1 element to several 16 K-element chunks) == the per-variable call, bit for bit."""
    rng = np.random.default_rng(10)
    sizes = [1, 7, 256, 16384, 16385, 70001, 3 * 17 * 33]
    mk = lambda n, s: rng.normal(0, s, n).astype(np.float32)
    host = [(mk(n, 1.0), mk(n, 1.5), mk(n, 0.1), np.abs(mk(n, 0.1))) for n in sizes]
    a = [[dev(t) for t in h] for h in host]; b = [[dev(t) for t in h] for h in host]
    b[-1] = [t.view(3, 17, 33) for t in b[-1]]                         # any rank, as the variables of a network have
    for step in (1, 2, 4000):
        for var, g, m, v in a:
            ops.adam_step(var, g, m, v, step, 2e-4)
        ops.adam_step_multi([t[0] for t in b], [t[1] for t in b], [t[2] for t in b], [t[3] for t in b], step, 2e-4)
    torch.cuda.synchronize()
    for ta, tb in zip(a, b):
        for x, y in zip(ta, tb):
            assert torch.equal(x.reshape(-1), y.reshape(-1))
    with pytest.raises(ValueError):
        ops.adam_step_multi([a[0][0]], [a[1][1]], [a[0][2]], [a[0][3]], 1, 1e-3)


def test_adam_steps_vs_oracle(ops):
    from dc_tts_amd.train import learning_rate_decay
    rng = np.random.default_rng(9)
    n = 5000
    var = rng.normal(0, 1, n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    vr, mr, vvr = var.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    dv, dm, dvv = dev(var), dev(m), dev(v)
    for step in range(1, 6):
        g = (rng.normal(0, 1.5, n)).astype(np.float32)                 # some |g| > 1: exercises the clip
        lr = learning_rate_decay(0.001, step - 1)
        assert abs(lr - TR.learning_rate_decay(0.001, step - 1)) < 1e-18
        vr, mr, vvr = TR.adam_step(vr, g.astype(np.float64), mr, vvr, step, lr)
        ops.adam_step(dv, dev(g), dm, dvv, step, lr)
    torch.cuda.synchronize()
    # fp32 constants: 1 - 0.999f = 0.00100005 (4.7e-5 off), as in TensorFlow's fp32 Adam kernel; the oracle is float64
    assert np.abs(dv.cpu().numpy() - vr).max() < 1e-6 and rel(dm.cpu().numpy(), mr) < 1e-5 and rel(dvv.cpu().numpy(), vvr) < 1e-4


def _synthetic_batches(n, B=4, N=24, T=16, seed=3):
    from dc_tts_amd.hyperparams import hp
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        texts = rng.integers(2, len(hp.vocab), (B, N)).astype(np.int32); texts[:, -1] = 1
        out.append((texts, rng.random((B, T, hp.n_mels), dtype=np.float32), rng.random((B, 4 * T, hp.n_linear), dtype=np.float32), ["x.wav"] * B))
    return out


# First run on a GPU box in round 3 (profiles/r03_unverified_pass.txt): green.
@pytest.mark.parametrize("num", [1, 2])
def test_training_loop_checkpoints_and_resumes(tmp_path, num):
    """dc_tts_amd.train.main = train.py:137-162: a run of 6 steps, and a run of 4 steps that is stopped and resumed from its checkpoint
    for 2 more (variables, Adam slots and global_step restored, as the Supervisor does), end in the same checkpoint, bit for bit."""
    from dc_tts_amd import train as TRN
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.tf_checkpoint import latest_checkpoint, read_checkpoint
    bt = _synthetic_batches(8)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    assert TRN.main([str(num), "--logdir", a, "--num-iterations", "5"], save_every=2, batches=iter(bt)) == 0       # steps 1..6, saves at 2, 4, 6
    assert TRN.main([str(num), "--logdir", b, "--num-iterations", "3"], save_every=2, batches=iter(bt)) == 0       # steps 1..4
    mid = read_checkpoint(latest_checkpoint(b + f"-{num}"), verify_tensors=False)
    assert int(mid["gs/global_step"]) == 4
    assert TRN.main([str(num), "--logdir", b, "--num-iterations", "5"], save_every=2, batches=iter(bt[4:])) == 0   # resumed: steps 5, 6
    A, Bc = read_checkpoint(latest_checkpoint(a + f"-{num}"), verify_tensors=False), read_checkpoint(latest_checkpoint(b + f"-{num}"))
    assert int(A["gs/global_step"]) == int(Bc["gs/global_step"]) == 6 and set(A) == set(Bc)
    prefix = "Text2Mel/" if num == 1 else "SSRN/"
    names = [n for n in A if n.startswith(prefix)]
    assert len(names) == 3 * (209 if num == 1 else 80)                       # variables + Adam + Adam_1
    moved = 0
    for n in names:
        np.testing.assert_array_equal(A[n], Bc[n], err_msg=n)
        if not n.endswith(("Adam", "Adam_1")):
            moved += int(np.abs(A[n] - mid[n]).max() > 0)
    assert moved > len(names) // 3 * 0.9                                     # the resumed steps did train
    assert (num == 2) or os.path.exists(os.path.join(b + "-1", "alignment_000k.png")) or os.path.exists(os.path.join(b + "-1", "alignment_000k.npy"))


def test_training_loop_on_a_wave_corpus(tmp_path):
    """wave files -> prepo -> bucketed batches -> two Text2Mel steps and two SSRN steps, end to end."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_train_input import _corpus
    from dc_tts_amd import prepo, train as TRN
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.tf_checkpoint import latest_checkpoint, read_checkpoint
    root, rows = _corpus(tmp_path, n=12)
    h = hp.replace(data=root, B=4, logdir=str(tmp_path / "log"))
    pre = str(tmp_path / "pre")
    assert prepo.main(["--out", pre], hp=h) == 0
    for num in (1, 2):
        assert TRN.main([str(num), "--prepro-dir", pre, "--num-iterations", "1"], hp=h, save_every=2) == 0
        t = read_checkpoint(latest_checkpoint(h.logdir + f"-{num}"))
        assert int(t["gs/global_step"]) == 2 and all(np.isfinite(v).all() for v in t.values())
