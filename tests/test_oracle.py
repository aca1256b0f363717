"""Known-answer and structural tests that pin the CPU oracle (oracle/dctts_ref.py).

The reference has no tests or golden vectors, so these are authored from the TF
semantics listed in SURVEY Appendix B, plus an independent cross-check of every primitive against
torch.nn.functional, plus the committed golden fixtures (tests/golden/).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dc_tts_amd.hyperparams import hp
from dc_tts_amd.layers import audiodec_cone, variable_shapes
from oracle import dctts_ref as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- KATs
def test_layernorm_kat():
    x = np.array([[1., 2., 3., 4.]], np.float32)
    y = O.normalize(x, np.ones(4, np.float32), np.zeros(4, np.float32))
    np.testing.assert_allclose(y[0], [-1.3416408, -0.4472136, 0.4472136, 1.3416408], atol=1e-6)
    z = O.normalize(np.full((1, 8), 3.25, np.float32), np.ones(8, np.float32), np.zeros(8, np.float32))
    assert np.all(z == 0) and not np.isnan(z).any()          # all-equal row -> exactly 0 (eps 1e-12)


def test_embed_row0_is_zero():
    tab = np.arange(32 * 4, dtype=np.float32).reshape(32, 4) + 1
    out = O.embed(np.array([[0, 1, 0, 5]]), tab)
    assert np.all(out[0, 0] == 0) and np.all(out[0, 2] == 0)
    np.testing.assert_array_equal(out[0, 1], tab[1]); np.testing.assert_array_equal(out[0, 3], tab[5])


def test_deconv_phase_kat():
    """x=[1,1], W=[1,1,1], no bias -> [1,1,2,1] before LN."""
    P = {"D/conv2d_transpose/kernel": np.ones((1, 3, 1, 1), np.float32),
         "D/conv2d_transpose/bias": np.zeros(1, np.float32),
         "D/normalize/gamma": np.ones(1, np.float32), "D/normalize/beta": np.zeros(1, np.float32)}
    x = np.ones((1, 2, 1), np.float32)
    W = P["D/conv2d_transpose/kernel"][0]
    xm1 = np.pad(x, ((0, 0), (1, 0), (0, 0)))[:, :2]
    y = np.zeros((1, 4, 1), np.float32)
    y[:, 0::2] = x @ W[0].T + xm1 @ W[2].T
    y[:, 1::2] = x @ W[1].T
    np.testing.assert_array_equal(y[0, :, 0], [1, 1, 2, 1])


def test_causality(weights):
    """Changing S[:, t0:] leaves Q[:, :t0] bit-identical (SURVEY B.2 KAT)."""
    rng = np.random.default_rng(0)
    S = rng.random((1, 40, hp.n_mels), dtype=np.float32)
    S2 = S.copy(); S2[:, 25:] = rng.random((1, 15, hp.n_mels), dtype=np.float32)
    Q1 = O.AudioEnc(S, weights, hp); Q2 = O.AudioEnc(S2, weights, hp)
    np.testing.assert_array_equal(Q1[:, :25], Q2[:, :25])
    assert np.abs(Q1[:, 25:] - Q2[:, 25:]).max() > 1e-3


@pytest.mark.parametrize("p", [0, 1, 176, 177, 178, 179])
def test_attention_window(p):
    """Allowed keys are p <= n < min(p+3, 180); softmax over them sums to 1; everything else exactly 0."""
    h = hp.replace(max_T=4)
    rng = np.random.default_rng(p)
    Q = rng.standard_normal((1, 4, h.d)).astype(np.float32)
    K = rng.standard_normal((1, h.max_N, h.d)).astype(np.float32)
    V = rng.standard_normal((1, h.max_N, h.d)).astype(np.float32)
    R, al, mx = O.Attention(Q, K, V, h, True, np.array([p], np.int32))
    A = al.transpose(0, 2, 1)[0]
    lo, hi = p, min(p + 3, h.max_N)
    assert np.all(A[:, :lo] == 0) and np.all(A[:, hi:] == 0)
    np.testing.assert_allclose(A[:, lo:hi].sum(-1), 1.0, atol=1e-6)
    assert np.all((mx[0] >= lo) & (mx[0] < hi)) and mx.dtype == np.int64
    assert R.shape == (1, 4, 2 * h.d) and al.shape == (1, h.max_N, 4)
    np.testing.assert_array_equal(R[..., h.d:], Q)


def test_attention_tie_lowest_index():
    h = hp.replace(max_T=1)
    Q = np.zeros((1, 1, h.d), np.float32)            # all logits equal -> tie inside the window
    K = np.ones((1, h.max_N, h.d), np.float32); V = K.copy()
    _, _, mx = O.Attention(Q, K, V, h, True, np.array([7], np.int32))
    assert mx[0, 0] == 7


def test_mask_constant():
    assert O.NEG == -4294967295.0 and np.float32(O.NEG) == np.float32(-4294967296.0)


def test_counts_match_survey(weights):
    assert sum(v.size for v in weights.values()) == 52_380_671
    shapes = variable_shapes(hp)
    t2m = sum(int(np.prod(s)) for n, s in shapes.items() if n.startswith("Text2Mel"))
    assert t2m == 23_970_288 and len(shapes) == len(weights)
    assert [len(c) for c in audiodec_cone(hp)] == [85, 83, 45, 15, 5, 3, 1, 1, 1, 1, 1]


def test_text_front_end():
    """Config 1 input: Harvard sentence 1 -> 'the birch canoe slid on the smooth planks.E' (data_load.py:79-86)."""
    L = O.load_sentences(["1. The birch canoe slid on the smooth planks.\n"], hp)
    s = "the birch canoe slid on the smooth planks.E"
    assert L.shape == (1, 180) and L.dtype == np.int32
    assert "".join(hp.vocab[i] for i in L[0, :len(s)]) == s and np.all(L[0, len(s):] == 0)


# ---------------------------------------------------------------- independent cross-check vs torch.nn.functional
def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize("k,rate,padding", [(3, 1, "SAME"), (3, 9, "SAME"), (3, 27, "CAUSAL"), (3, 3, "CAUSAL"), (1, 1, "SAME")])
def test_conv_vs_torch(k, rate, padding):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 50, 16)); W = rng.standard_normal((k, 16, 24)); b = rng.standard_normal(24)
    y = O._conv(x, W, b, rate, padding)
    xt = _t(x).transpose(1, 2)
    total = (k - 1) * rate
    pl, pr = (total, 0) if padding == "CAUSAL" else (total // 2, total - total // 2)
    yt = F.conv1d(F.pad(xt, (pl, pr)), _t(W).permute(2, 1, 0).contiguous(), _t(b), dilation=rate).transpose(1, 2)
    np.testing.assert_allclose(y, yt.numpy(), atol=1e-10)


def test_layernorm_vs_torch():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 7, 33)); g = rng.standard_normal(33); b = rng.standard_normal(33)
    y = O.normalize(x, g, b)
    yt = F.layer_norm(_t(x), (33,), _t(g), _t(b), eps=1e-12)
    np.testing.assert_allclose(y, yt.numpy(), atol=1e-10)


def test_deconv_vs_torch():
    """conv_transpose1d(stride=2, padding=0)[..., :-1] is the TF 'same' stride-2 k=3 transpose."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 9, 6)); Wt = rng.standard_normal((1, 3, 5, 6)); b = rng.standard_normal(5)
    P = {"D/conv2d_transpose/kernel": Wt, "D/conv2d_transpose/bias": b,
         "D/normalize/gamma": np.ones(5), "D/normalize/beta": np.zeros(5)}
    y = O.conv1d_transpose(x, P, "D")
    w = _t(Wt[0]).permute(2, 1, 0).contiguous()        # (Cin, Cout, k)
    yt = F.conv_transpose1d(_t(x).transpose(1, 2), w, _t(b), stride=2)[..., :-1].transpose(1, 2)
    yt = F.layer_norm(yt, (5,), eps=1e-12)
    assert y.shape == (2, 18, 5)
    np.testing.assert_allclose(y, yt.numpy(), atol=1e-9)


# ---------------------------------------------------------------- the incremental algorithm == the reference loop
def test_incremental_model_equals_reference_loop(weights):
    """The decode the HIP path implements (cached AudioEnc + 85-row cone with the current window) reproduces the
    restated full-recompute loop; the frozen-R cache does not (SURVEY B.7 regression guard).  T=100 > 85."""
    from oracle.incremental_ref import incremental_decode
    from dc_tts_amd.weights import synthetic_text
    h = hp.replace(max_T=100)
    L = synthetic_text(h, B=2, seed=7)
    Y, _, traj = O.synthesize(L, weights, h, np.float64, run_ssrn=False)
    Yi, traji = incremental_decode(L, weights, h, np.float64)
    np.testing.assert_array_equal(traj, traji)
    assert np.abs(Y - Yi).max() < 1e-9
    assert traj.max() > 10                                   # attention really moved
    Yf, trajf = incremental_decode(L, weights, h, np.float64, frozen_R=True)
    assert np.abs(Y - Yf).max() > 1e-2                       # the 'obvious' cache is a different function


def short_text(h, B, seed):
    """(B, h.max_N) ids for a tiny max_N (the attention window must run into the end of the text within a short decode)."""
    rng = np.random.default_rng(seed)
    L = rng.integers(2, len(h.vocab), (B, h.max_N)).astype(np.int32)
    L[:, -1] = 1
    return L


def test_v3_model_equals_reference_loop(weights):
    """Round-2 decode data flow (oracle/incremental_ref.incremental_decode_v3: AudioDec C_1 as a row operation on V.W_top / Q.W_bot,
    the two older taps of every causal k=3 layer as presums) == the restated synthesize.py loop, fp64: the reorganisation is
    exact algebra, not an approximation."""
    from oracle.incremental_ref import incremental_decode_v3
    from dc_tts_amd.weights import synthetic_text
    h = hp.replace(max_T=100)
    L = synthetic_text(h, B=2, seed=7)
    Y, _, traj = O.synthesize(L, weights, h, np.float64, run_ssrn=False)
    st = {}
    Y3, traj3 = incremental_decode_v3(L, weights, h, np.float64, stats=st)
    np.testing.assert_array_equal(traj, traj3)
    assert np.abs(Y - Y3).max() < 1e-9
    assert st["min_top2_logit_gap"] > 0


def test_end_of_text_window_clipping_models(weights):
    """networks.py:142-147 once prev_max >= max_N - 2: the window is clipped to 2, then 1 key.  A 10-character text saturates
    within ~40 frames; both incremental models must follow the restated loop through and beyond saturation (fp32, trajectory
    integer-exact)."""
    from oracle.incremental_ref import incremental_decode, incremental_decode_v3
    h = hp.replace(max_N=10, max_T=90)
    L = short_text(h, 2, 11)
    Y, _, traj = O.synthesize(L, weights, h, np.float32, run_ssrn=False)
    assert traj.max() == h.max_N - 1 and (traj[:, -1] == h.max_N - 1).all()
    assert (traj == h.max_N - 2).any()                       # the 2-key window was visited too
    for fn in (incremental_decode, incremental_decode_v3):
        Yi, ti = fn(L, weights, h, np.float32)
        np.testing.assert_array_equal(traj, ti)
        assert np.abs(Y - Yi).max() < 1e-4


# ---------------------------------------------------------------- golden fixtures
def _gold(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def test_golden_networks(weights):
    g = _gold("networks_seed1234.npz")
    L = g["L"]; S = g["S"]
    K, V = O.TextEnc(L, weights, hp)
    np.testing.assert_allclose(K[:, ::9, ::8], g["K_sub"], atol=2e-5)
    np.testing.assert_allclose(V[:, ::9, ::8], g["V_sub"], atol=2e-5)
    Q = O.AudioEnc(S, weights, hp)
    np.testing.assert_allclose(Q[:, ::3, ::4], g["Q_sub"], atol=2e-5)
    h = hp.replace(max_T=S.shape[1])
    R, al, mx = O.Attention(Q, K, V, h, True, g["prev_max"])
    np.testing.assert_array_equal(mx, g["max_att"])
    lg, Y = O.AudioDec(R, weights, hp)
    np.testing.assert_allclose(Y, g["Y"], atol=2e-5)
    zl, Z = O.SSRN(Y[:, :8], weights, hp)
    np.testing.assert_allclose(Z[:, :, ::16], g["Z_sub"], atol=2e-5)


def test_golden_config1_loop(weights):
    """Config 1 of BASELINE.json: one Harvard sentence through the restated synthesize.py loop."""
    g = _gold("config1_harvard1.npz")
    h = hp.replace(max_T=int(g["max_T"]))
    Y, Z, traj = O.synthesize(g["L"], weights, h, np.float32, run_ssrn=False)
    np.testing.assert_array_equal(traj, g["traj"])
    np.testing.assert_allclose(Y, g["Y"], atol=5e-5)


def test_torch_restatement_matches_the_numpy_oracle():
    """oracle/torch_ref.py (what bench.py's cpu_baseline leg times on all host cores) is the same arithmetic as the
    numpy oracle: a few steps of the full-graph loop and one SSRN pass agree to fp32 re-association, trajectory integer-exact."""
    import numpy as np
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.weights import synthetic_text, synthetic_weights
    from oracle import dctts_ref as O
    from oracle import torch_ref as TR
    h = hp.replace(max_T=6)
    W = synthetic_weights(h, seed=1234, perturb=True)
    L = synthetic_text(h, B=2, seed=3)
    Yn, Zn, tn = O.synthesize(L, W, h, np.float32)
    Yt, Zt, tt = TR.synthesize(L, W, h)
    np.testing.assert_array_equal(tn, tt)
    assert np.abs(Yn - Yt).max() < 2e-5 and np.abs(Zn - Zt).max() < 2e-4
