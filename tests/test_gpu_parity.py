"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerance: BASELINE.json's north_star asks for mel / linear spectrograms within 1e-3 max-abs fp32 of the
reference; the oracle is the reference restatement.  Integer outputs (attention trajectory) must be exact.
Run on the GPU box with  python -m pytest tests -m gpu.
"""
import os

import numpy as np
import pytest
import torch

from dc_tts_amd.hyperparams import hp
from dc_tts_amd.layers import audiodec_layers, audioenc_layers, ssrn_layers, textenc_layers
from dc_tts_amd.weights import synthetic_text
from oracle import dctts_ref as O

pytestmark = pytest.mark.gpu

TOL = 1e-3          # north_star: max-abs fp32 on the (sigmoid) spectrogram outputs
TOL_INNER = 2e-3    # un-squashed intermediate activations (|x| up to ~5): same relative class
GOLD = os.path.join(os.path.dirname(__file__), "golden")

_engines = {}
DEFAULT_MODE = 3    # decode v3 (hoisted taps); 1 / 2 = round-1 split kernels, 0 = fused full-row kernels


def engine_for(weights, max_T=hp.max_T, max_N=hp.max_N):
    from dc_tts_amd.engine import Engine
    key = (max_T, max_N)
    if key not in _engines:
        _engines[key] = Engine(weights, hp.replace(max_T=max_T, max_N=max_N))
    return _engines[key]


def top2_gap_ulps(traj_trace):
    """Smallest gap between the two largest window logits of any newest-row decision, in ulps of the larger (SURVEY section 7:
    the arg-max is fed back, so a decision closer than the fp32 re-association noise of the HIP path could flip the trajectory)."""
    return min(traj_trace) if traj_trace else float("inf")


def oracle_loop_with_margin(L, weights, h):
    """O.synthesize plus the minimum top-2 gap (in fp32 ulps) of the step-j row's allowed logits."""
    gaps = []

    def trace(j, g):
        A = g["alignments"][:, :, j]                       # (B, N) post-softmax row j
        for b in range(A.shape[0]):
            nz = np.flatnonzero(A[b] > 0)
            if len(nz) < 2:
                continue
            q = g["Q"][b, j].astype(np.float64)
            lg = np.sort((g["K"][b, nz].astype(np.float64) @ q) / 16.0)
            top = np.float32(lg[-1])
            gaps.append(float((lg[-1] - lg[-2]) / np.spacing(np.abs(top))))
    Y, _, traj = O.synthesize(L, weights, h, np.float32, run_ssrn=False, trace=trace)
    return Y, traj, top2_gap_ulps(gaps)


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_native_library_is_loaded(weights):
    """The product path runs in libdctts_hip.so, not in a torch fallback."""
    eng = engine_for(weights)
    assert eng.lib._name.endswith("dc_tts_amd/lib/libdctts_hip.so")
    with open("/proc/self/maps") as f:
        assert "libdctts_hip.so" in f.read()
    assert eng.device_bytes() > 200e6        # 209.5 MB of fp32 weights packed on the device


# ---------------------------------------------------------------- every kernel shape class, one layer at a time
def _dev_index(layers, li, net):
    """logical layer index -> device layer index (textenc: embed is fused into C_2; ssrn: D = 2 entries)."""
    if net == "textenc":
        return li - 1
    if net == "ssrn":
        return li + sum(1 for l in layers[:li] if l.kind == "D")
    return li


CASES = []
for net, fn, scope, causal in (("textenc", textenc_layers, "Text2Mel/TextEnc", False), ("audioenc", audioenc_layers, "Text2Mel/AudioEnc", True),
                               ("audiodec", audiodec_layers, "Text2Mel/AudioDec", True), ("ssrn", ssrn_layers, "SSRN", False)):
    seen = set()
    for li, l in enumerate(fn(hp)):
        key = (l.kind, l.cin, l.cout, l.size, l.rate, l.act)
        if l.kind == "E" or key in seen:
            continue
        seen.add(key)
        CASES.append(pytest.param(net, scope, causal, li, id=f"{net}-{l.scope}-{l.kind}{l.cin}to{l.cout}k{l.size}d{l.rate}"))


@pytest.mark.parametrize("net,scope,causal,li", CASES)
def test_layer_vs_oracle(weights, net, scope, causal, li):
    fn = {"textenc": textenc_layers, "audioenc": audioenc_layers, "audiodec": audiodec_layers, "ssrn": ssrn_layers}[net]
    layers = fn(hp)
    l = layers[li]
    eng = engine_for(weights)
    rng = np.random.default_rng(100 + li)
    B, T = 2, 75                                   # 150 rows: 4 full 32-row tiles + a ragged one
    P = O._Scoped(weights, scope, np.float32)
    pad = "CAUSAL" if causal else "SAME"
    di = _dev_index(layers, li, net)
    if net == "textenc" and li == 1:               # embed + C_2 fused
        ids = rng.integers(0, len(hp.vocab), (B, T)).astype(np.int32)
        ref = O.conv1d(O.embed(ids, P["embed_1/lookup_table"]), P, "C_2", act=O.relu)
        got = eng.debug_layer(net, 0, dev(ids), l.cout).cpu().numpy()
    else:
        x = rng.standard_normal((B, T, l.cin)).astype(np.float32)
        if l.kind == "C":
            ref = O.conv1d(x, P, l.scope, padding=pad, act=O.relu if l.act == "relu" else None)
            got = eng.debug_layer(net, di, dev(x), l.cout).cpu().numpy()
        elif l.kind == "HC":
            ref = O.hc(x, P, l.scope, rate=l.rate, padding=pad)
            got = eng.debug_layer(net, di, dev(x), l.cout).cpu().numpy()
        else:
            ref = O.conv1d_transpose(x, P, l.scope)
            got = eng.debug_layer(net, di, dev(x), l.cout, upsample=2).cpu().numpy()
    # the last layer of AudioDec / SSRN carries the network's sigmoid in its epilogue
    if (net == "audiodec" and li == len(layers) - 1) or (net == "ssrn" and li == len(layers) - 1):
        ref = O.sigmoid(ref)
    assert got.shape == ref.shape
    err = maxabs(got, ref)
    assert err < 2e-4, f"{net}/{l.scope}: max-abs {err}"


# SSRN layers at 4T resolution are launched as exact rounds of 32-row items (hconv_kernel) + a tail of 16-row items
# (hconv16_kernel).  The small cases above land entirely on the 16-row kernel; this case has one exact round for the
# 32-row kernel plus a ragged tail (40 rows = 2 full 16-row items + one of 8 rows) for the other.
SPLIT_CASES = []
_seen = set()
_ssrn = ssrn_layers(hp)
_first4t = [i for i, l in enumerate(_ssrn) if l.kind == "D"][1] + 1
for li in range(_first4t, len(_ssrn)):
    l = _ssrn[li]
    key = (l.kind, l.cin, l.cout, l.act)
    if key not in _seen:
        _seen.add(key)
        SPLIT_CASES.append(pytest.param(li, id=f"ssrn-{l.scope}-{l.kind}{l.cin}to{l.cout}"))


@pytest.mark.parametrize("li", SPLIT_CASES)
def test_layer_row_split(weights, li):
    l = _ssrn[li]
    eng = engine_for(weights)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    B, T = 2, n_cu * 16 + 20                        # rows = n_cu * 32 + 40
    rng = np.random.default_rng(300 + li)
    P = O._Scoped(weights, "SSRN", np.float32)
    x = rng.standard_normal((B, T, l.cin)).astype(np.float32)
    if l.kind == "C":
        ref = O.conv1d(x, P, l.scope, padding="SAME", act=O.relu if l.act == "relu" else None)
    else:
        ref = O.hc(x, P, l.scope, rate=l.rate, padding="SAME")
    if li == len(_ssrn) - 1:
        ref = O.sigmoid(ref)
    got = eng.debug_layer("ssrn", _dev_index(_ssrn, li, "ssrn"), dev(x), l.cout).cpu().numpy()
    assert got.shape == ref.shape
    # worst rows of either part, reported separately so a failure names the kernel
    flat_g, flat_r = got.reshape(B * T, -1), ref.reshape(B * T, -1)
    e32 = maxabs(flat_g[: n_cu * 32], flat_r[: n_cu * 32]); e16 = maxabs(flat_g[n_cu * 32:], flat_r[n_cu * 32:])
    assert e32 < 2e-4 and e16 < 2e-4, f"ssrn/{l.scope}: max-abs 32-row part {e32}, 16-row tail {e16}"


# ---------------------------------------------------------------- network functions (the drop-in boundary)
def test_textenc(weights):
    eng = engine_for(weights)
    L = synthetic_text(hp, B=3, seed=5)
    K, V = eng.text_enc(dev(L))
    Kr, Vr = O.TextEnc(L, weights, hp)
    assert maxabs(K.cpu().numpy(), Kr) < TOL_INNER and maxabs(V.cpu().numpy(), Vr) < TOL_INNER


def test_textenc_out_of_vocabulary_ids_read_the_pad_row(weights):
    """Character ids outside the 32-entry table must not read out of bounds: they behave like PAD (row 0 = zeros), which is
    what tf.nn.embedding_lookup does on a GPU (on a CPU it raises)."""
    eng = engine_for(weights)
    L = synthetic_text(hp, B=2, seed=6)
    Lbad = L.copy(); Lbad[0, 3] = 77; Lbad[1, 10] = -5
    Lpad = L.copy(); Lpad[0, 3] = 0; Lpad[1, 10] = 0
    K1, V1 = eng.text_enc(dev(Lbad))
    K2, V2 = eng.text_enc(dev(Lpad))
    assert torch.equal(K1, K2) and torch.equal(V1, V2)


def test_audioenc_and_causality(weights):
    eng = engine_for(weights)
    rng = np.random.default_rng(6)
    S = rng.random((2, 70, hp.n_mels), dtype=np.float32)
    Q = eng.audio_enc(dev(S)).cpu().numpy()
    assert maxabs(Q, O.AudioEnc(S, weights, hp)) < TOL_INNER
    S2 = S.copy(); S2[:, 40:] = rng.random((2, 30, hp.n_mels), dtype=np.float32)
    Q2 = eng.audio_enc(dev(S2)).cpu().numpy()
    np.testing.assert_array_equal(Q[:, :40], Q2[:, :40])          # bit-identical prefix: causal


@pytest.mark.parametrize("mono", [True, False])
def test_attention(weights, mono):
    T = 40
    eng = engine_for(weights, max_T=T)
    h = hp.replace(max_T=T)
    rng = np.random.default_rng(7)
    Q = rng.standard_normal((3, T, h.d)).astype(np.float32)
    K = rng.standard_normal((3, h.max_N, h.d)).astype(np.float32)
    V = rng.standard_normal((3, h.max_N, h.d)).astype(np.float32)
    pm = np.array([0, 57, 178], np.int32)
    R, al, mx = eng.attention(dev(Q), dev(K), dev(V), mono, dev(pm) if mono else None)
    Rr, alr, mxr = O.Attention(Q, K, V, h, mono, pm if mono else None)
    assert mx.dtype == torch.int64 and tuple(al.shape) == (3, h.max_N, T)
    np.testing.assert_array_equal(mx.cpu().numpy(), mxr)
    assert maxabs(R.cpu().numpy(), Rr) < 1e-4 and maxabs(al.cpu().numpy(), alr) < 1e-5
    if mono:
        a = al.cpu().numpy()
        assert np.all(a[1, :57] == 0) and np.all(a[1, 60:] == 0)   # exact zeros outside the window


def test_attention_errors(weights):
    eng = engine_for(weights, max_T=40)
    Q = torch.zeros(1, 40, hp.d, device="cuda"); K = torch.zeros(1, 100, hp.d, device="cuda")
    with pytest.raises(ValueError):            # monotonic mask is built from hp.max_N (networks.py:142)
        eng.attention(Q, K, K, True, torch.zeros(1, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        eng.attention(Q, K, K, True, None)


def test_networks_training_true_is_the_reference_default(weights):
    """networks.py:14,73,157,214 default to training=True (dropout behind every block, modules.py:139,195,245).  The boundary functions honour it with
    the training forward of include/dctts_train.h: the result is the oracle's forward pass with the restated dropout masks (oracle/train_ref.py:
    TensorFlow's random stream cannot be reproduced), differs from the training=False result, and hp.dropout_rate = 0 makes the two forms agree."""
    from dc_tts_amd import networks
    from dc_tts_amd.layers import audiodec_layers, audioenc_layers
    from oracle import train_ref as TR
    T = 12
    eng = engine_for(weights, max_T=T)
    networks.bind(eng)
    rng = np.random.default_rng(3)
    S = rng.random((2, T, hp.n_mels), dtype=np.float32)
    networks.set_training_seed(7)
    Qt = networks.AudioEnc(dev(S))                                                      # the reference's default: training=True (call number 0)
    Qi = networks.AudioEnc(dev(S), training=False)
    W64 = {n: v.astype(np.float64) for n, v in weights.items() if n.startswith("Text2Mel/AudioEnc")}
    Qr, _ = TR.network_forward(audioenc_layers(hp), W64, "Text2Mel/AudioEnc", S.astype(np.float64), "causal", (hp.dropout_rate, 7, 0))
    assert maxabs(Qt.cpu().numpy(), Qr) < TOL_INNER
    assert float((Qt - Qi).abs().max()) > 1e-2                                          # dropout really happened
    R = rng.standard_normal((2, T, 2 * hp.d)).astype(np.float32)
    lg, Y = networks.AudioDec(dev(R))                                                   # call number 1
    Wd = {n: v.astype(np.float64) for n, v in weights.items() if n.startswith("Text2Mel/AudioDec")}
    lr, _ = TR.network_forward(audiodec_layers(hp), Wd, "Text2Mel/AudioDec", R.astype(np.float64), "causal", (hp.dropout_rate, 7, 1))
    assert maxabs(lg.cpu().numpy(), lr) < TOL_INNER and maxabs(Y.cpu().numpy(), 1.0 / (1.0 + np.exp(-lr))) < TOL
    K, V = networks.TextEnc(dev(synthetic_text(hp, B=2, seed=5)))
    _, Z = networks.SSRN(Y[:, :4].contiguous())
    assert tuple(K.shape) == (2, hp.max_N, hp.d) == tuple(V.shape) and tuple(Z.shape) == (2, 16, hp.n_linear) and bool(torch.isfinite(Z).all())


def test_audiodec(weights):
    eng = engine_for(weights)
    rng = np.random.default_rng(8)
    R = rng.standard_normal((2, 100, 2 * hp.d)).astype(np.float32)
    lg, Y = eng.audio_dec(dev(R))
    lgr, Yr = O.AudioDec(R, weights, hp)
    assert maxabs(Y.cpu().numpy(), Yr) < TOL and maxabs(lg.cpu().numpy(), lgr) < TOL_INNER


def test_ssrn(weights):
    eng = engine_for(weights)
    rng = np.random.default_rng(9)
    Y = rng.random((2, 37, hp.n_mels), dtype=np.float32)          # odd length: ragged tiles at T, 2T and 4T
    lg, Z = eng.ssrn(dev(Y))
    lgr, Zr = O.SSRN(Y, weights, hp)
    assert tuple(Z.shape) == (2, 148, hp.n_linear)
    assert maxabs(Z.cpu().numpy(), Zr) < TOL and maxabs(lg.cpu().numpy(), lgr) < 5e-3


def test_ssrn_only_batch128_config3(weights):
    """BASELINE configs[2]: SSRN-only, batch 128, (128, 210, 80) -> (128, 840, 1025).  Determinism, agreement with the same
    utterances run as shards of 32 (rows land on the 32-row or the 16-row MFMA kernel depending on the launch: fp32
    reassociation, <= 1e-5), range, and the oracle on two utterances."""
    eng = engine_for(weights)
    rng = np.random.default_rng(128)
    Yh = rng.random((128, hp.max_T, hp.n_mels), dtype=np.float32)
    Y = dev(Yh)
    Z = eng.ssrn(Y, want_logits=False)[1]
    Z2 = eng.ssrn(Y, want_logits=False)[1]
    assert tuple(Z.shape) == (128, 4 * hp.max_T, hp.n_linear) and torch.equal(Z, Z2)
    for s0 in (0, 96):
        Zs = eng.ssrn(Y[s0:s0 + 32].contiguous(), want_logits=False)[1]
        assert float((Zs - Z[s0:s0 + 32]).abs().max()) < 1e-5
    assert float(Z.min()) >= 0.0 and float(Z.max()) <= 1.0 and bool(torch.isfinite(Z).all())
    idx = list(range(3, 128, 8))                                 # 16 of the 128 utterances, spread over every shard of 32 and both kernel forms
    _, Zr = O.SSRN(Yh[idx], weights, hp)
    e = maxabs(Z[idx].cpu().numpy(), Zr)
    print(f"config 3: max|Z - oracle| over {len(idx)} utterances = {e:.2e}")
    assert e < TOL


def test_networks_surface_and_golden(weights):
    """The reference-named functions on the committed golden inputs (tests/golden/networks_seed1234.npz)."""
    from dc_tts_amd import networks
    g = np.load(os.path.join(GOLD, "networks_seed1234.npz"))
    T = g["S"].shape[1]
    networks.bind(engine_for(weights, max_T=T))
    K, V = networks.TextEnc(dev(g["L"]), training=False)
    Q = networks.AudioEnc(dev(g["S"]), training=False)
    R, al, mx = networks.Attention(Q, K, V, mononotic_attention=True, prev_max_attentions=dev(g["prev_max"]))
    lg, Y = networks.AudioDec(R, training=False)
    zl, Z = networks.SSRN(Y[:, :8].contiguous(), training=False)
    assert maxabs(K.cpu().numpy()[:, ::9, ::8], g["K_sub"]) < TOL_INNER
    assert maxabs(Q.cpu().numpy()[:, ::3, ::4], g["Q_sub"]) < TOL_INNER
    np.testing.assert_array_equal(mx.cpu().numpy(), g["max_att"])
    assert maxabs(Y.cpu().numpy(), g["Y"]) < TOL
    assert maxabs(Z.cpu().numpy()[:, :, ::16], g["Z_sub"]) < TOL


# ---------------------------------------------------------------- the autoregressive loop
_oracle_cache = {}


def _oracle_decode(weights, T, B, seed):
    key = (T, B, seed)
    if key not in _oracle_cache:
        h = hp.replace(max_T=T)
        L = synthetic_text(h, B=B, seed=seed)
        _oracle_cache[key] = (L,) + oracle_loop_with_margin(L, weights, h)
    return _oracle_cache[key]


@pytest.mark.parametrize("mode", [3, 0])
@pytest.mark.parametrize("graph", [0, 1])
def test_decode_vs_oracle_loop(weights, graph, mode):
    """Incremental exact decode == restated synthesize.py loop: integer-exact attention trajectory, Y within 1e-3.
    T = 100 > 85 so the full AudioDec dependency cone is exercised.  The oracle's closest arg-max decision is reported and
    must be far outside fp32 re-association noise (otherwise an exact trajectory would be luck)."""
    T = 100
    eng = engine_for(weights, max_T=T)
    eng.set_decode_graph(graph)
    eng.set_decode_mode(mode)
    L, Yr, trajr, gap = _oracle_decode(weights, T, 3, 21)
    Y, mx = eng.text2mel(dev(L))
    eng.set_decode_mode(DEFAULT_MODE)
    print(f"min top-2 window-logit gap of the oracle's decisions: {gap:.0f} ulp")
    assert gap > 100, f"closest arg-max decision only {gap} ulp apart: pick another seed or check against the fp64 oracle"
    np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
    err = maxabs(Y.cpu().numpy(), Yr)
    assert err < TOL, f"decode max-abs {err}"
    assert trajr.max() > 10


@pytest.mark.parametrize("knob", ["DCTTS_SYNC_VALUES=0", "DCTTS_CHAIN_WAIT=0", "DCTTS_XGROUP=0", "DCTTS_XCONE=0", "DCTTS_XCONE=2", "DCTTS_XGROUP=0,DCTTS_XCONE=0", "DCTTS_CHAIN_TAIL=0", "DCTTS_CHAIN_TAIL=1", "DCTTS_CHAIN_TAIL=5", "DCTTS_CHAIN_TAIL=6", "DCTTS_CHAIN_TAIL=6,DCTTS_XCONE=2"])
def test_decode_stream_meeting_variants(weights, knob):
    """The chain and side streams of the decode meet inside kernels (default: counters polled / written by the launches themselves, passenger
    workgroups), with stream wait / write operations (DCTTS_CHAIN_WAIT=0), or through events (DCTTS_SYNC_VALUES=0: what rocprofv3 --pmc
    needs); DCTTS_XGROUP=0 / DCTTS_XCONE=0 run the chain's / the side stream's highway layers as one launch per layer instead of the
    team kernels (the form a decode falls back to after a failed team hand-off); DCTTS_CHAIN_TAIL selects what follows the chain's AudioDec run:
    2 (default) the whole chain piece as ONE launch (xchain_kernel, round 5: xtail_kernel's layers -- the newest-row layers HC_2 .. HC_4, HC_5 .. HC_7 over their
    few cone rows, the seven k = 1 layers -- a team barrier, then xgroup_kernel's AudioEnc run + attention + C_1), 6 the same as two launches (round 4's form),
    5 xtail_kernel behind an AudioDec run of xgroup_kernel (three launches), 1 xmlp_kernel (the k = 1 layers in team form, HC_5 .. HC_7 split between chain and
    side stream), 0 mlp_rows_kernel (round 2's row-split form).  The knobs are read when a context is created.  Every
    variant must reproduce the oracle loop: trajectory integer-exact."""
    from dc_tts_amd.engine import Engine
    T = 100
    pairs = [kv.split("=") for kv in knob.split(",")]              # (both team kernels off = what a decode falls back to after a failed hand-off)
    old = {name: os.environ.get(name) for name, _ in pairs}
    for name, val in pairs: os.environ[name] = val
    try:
        eng = Engine(weights, hp.replace(max_T=T))
    finally:
        for name, _ in pairs:
            if old[name] is None: del os.environ[name]
            else: os.environ[name] = old[name]
    L, Yr, trajr, gap = _oracle_decode(weights, T, 3, 21)
    for graph in (0, 1):
        eng.set_decode_graph(graph)
        Y, mx = eng.text2mel(dev(L))
        eng.synchronize()
        np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
        assert maxabs(Y.cpu().numpy(), Yr) < TOL
    eng.close()


@pytest.mark.parametrize("B", [8, 7, 12, 16])
def test_small_batches_spread_over_all_teams_bitwise(weights, B):
    """Round 6: a batch of at most 8 / 16 utterances is dealt ONE / TWO to a team (XCD) and round instead of four, so that all eight teams work (BASELINE
    configs[4]'s share of a GPU is 8 utterances: rounds 3-5 ran it on two teams).  An utterance's arithmetic does not depend on the slot it sits in: mel
    frames and trajectory are bitwise those of the four-utterance form (DCTTS_XGROUP=2) and of the same utterances decoded inside a batch of 32."""
    from dc_tts_amd.engine import Engine
    T = 90
    h = hp.replace(max_T=T)
    Lh = synthetic_text(h, B=32, seed=77)
    os.environ["DCTTS_XGROUP"] = "2"
    try:
        e4 = Engine(weights, h)
    finally:
        del os.environ["DCTTS_XGROUP"]
    eu = Engine(weights, h)
    L = dev(Lh[:B])
    Y4, m4 = e4.text2mel(L); e4.synchronize()
    Yu, mu = eu.text2mel(L); eu.synchronize()
    assert torch.equal(Yu, Y4) and torch.equal(mu, m4)
    Y32, m32 = eu.text2mel(dev(Lh)); eu.synchronize()
    assert torch.equal(Y32[:B], Yu) and torch.equal(m32[:B], mu)
    from oracle.incremental_ref import incremental_decode_v3
    Yr, trajr = incremental_decode_v3(Lh[:B], weights, h, np.float32)
    np.testing.assert_array_equal(mu.cpu().numpy(), trajr)
    assert maxabs(Yu.cpu().numpy(), Yr) < TOL
    e4.close(); eu.close()


def test_first_decodes_at_new_geometries_with_fresh_inputs(weights):
    """What the other decode tests cannot see: they decode the SAME text several times at one geometry, so a kernel that reads a buffer before its producer has
    written it finds the previous decode's identical values there, and only the very first decode at a new geometry (new, zero-filled workspaces; new device
    tables; cold TLB) shows the race.  Round 6 had one (a weight-slice request inside a rolled loop of xcone_kernel's second row phase: a quarter of the first
    decodes wrong, every later one right; tools/flaky_probe.py).  Here every (B, T) is a new geometry, every decode a new text, each compared with the numpy
    statement of the incremental algorithm; a second engine is kept busy on the device in between."""
    from dc_tts_amd.engine import Engine
    from oracle.incremental_ref import incremental_decode_v3
    busy = engine_for(weights)
    Lb = dev(synthetic_text(hp, B=32, seed=5))
    bad = []
    for T in (60, 66, 72):
        h = hp.replace(max_T=T)
        eng = Engine(weights, h)
        for B in (5, 8, 32):
            for rep in range(2):                                       # the first decode at the geometry, then a DIFFERENT text at the same one
                Lh = synthetic_text(h, B=B, seed=1000 * T + 10 * B + rep)
                Yr, trajr = incremental_decode_v3(Lh, weights, h, np.float32)
                if rep: busy.ssrn(torch.rand(8, hp.max_T, hp.n_mels, device="cuda"))
                Y, mx = eng.text2mel(dev(Lh))
                eng.synchronize()
                e = maxabs(Y.cpu().numpy(), Yr)
                if e >= TOL or not np.array_equal(mx.cpu().numpy(), trajr):
                    bad.append((T, B, rep, e))
        eng.close()
    busy.text_enc(Lb); torch.cuda.synchronize()
    assert not bad, bad


def short_text(h, B, seed):
    rng = np.random.default_rng(seed)
    L = rng.integers(2, len(h.vocab), (B, h.max_N)).astype(np.int32)
    L[:, -1] = 1
    return L


@pytest.mark.parametrize("T", [1, 2, 3])
def test_decode_first_and_last_pieces_only(weights, T):
    """max_T = 1, 2, 3: a decode that consists of its first and last chain pieces (and at T = 3 ONE steady-state piece) -- the launches that differ from the
    steady state (round 5: the first piece is an xgroup_kernel launch, the last one an xtail_kernel launch without the AudioEnc run, the passengers' row-1
    presums come from a launch in front of the loop) -- at B = 1, 3 and 33 (a lone utterance, a team that is not full, a second round of utterance groups)."""
    h = hp.replace(max_T=T)
    eng = engine_for(weights, max_T=T)
    for B in (1, 3, 33):
        L = synthetic_text(h, B=B, seed=7)
        Yr, _, trajr = O.synthesize(L, weights, h, np.float32, run_ssrn=False)
        for graph in (0, 1):
            eng.set_decode_graph(graph)
            Y, mx = eng.text2mel(dev(L))
            eng.synchronize()
            np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
            err = maxabs(Y.cpu().numpy(), Yr)
            assert err < TOL, f"T {T} B {B} graph {graph}: decode max-abs {err}"


@pytest.mark.parametrize("mode", [3, 0])
def test_decode_end_of_text_window(weights, mode):
    """networks.py:142-147 at the end of the text: once prev_max >= max_N - 2 the window is clipped to 2, then 1 key.  A 10-character
    text saturates within ~40 frames; the decode must follow the restated loop through the 3 -> 2 -> 1 key regimes (the
    nk < win branch of every decode attention kernel, cone re-evaluation with 1-2 surviving keys), trajectory integer-exact."""
    h = hp.replace(max_N=10, max_T=90)
    eng = engine_for(weights, max_T=h.max_T, max_N=h.max_N)
    eng.set_decode_mode(mode)
    L = short_text(h, 5, 11)
    Yr, trajr, gap = oracle_loop_with_margin(L, weights, h)
    assert trajr.max() == h.max_N - 1 and (trajr[:, -1] == h.max_N - 1).all() and (trajr == h.max_N - 2).any()
    for graph in (0, 1):
        eng.set_decode_graph(graph)
        Y, mx = eng.text2mel(dev(L))
        np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
        err = maxabs(Y.cpu().numpy(), Yr)
        assert err < TOL, f"mode {mode} graph {graph}: decode max-abs {err}"
    eng.set_decode_mode(DEFAULT_MODE)
    assert gap > 100, gap


@pytest.mark.parametrize("mode", [3, 0])
def test_decode_end_of_text_at_production_geometry(weights, mode):
    """The end-of-text regimes of networks.py:142-147 at max_N = 180 (the production table geometry): random weights stall the attention near
    key 60, so the decode is SEEDED (dctts_debug_seed_prev_max, a test hook) with prev_max_attentions 170 .. 179 and compared with the oracle
    loop started the same way.  Covers the window on keys 177, 178, 179, its clipping to 2 keys (prev_max = 178) and 1 key (179) from the
    first frame on, and the last rows of the cached V.W / V.W.W tables the cone row operations index."""
    T = 30
    h = hp.replace(max_T=T)
    eng = engine_for(weights, max_T=T)
    eng.set_decode_mode(mode)
    prev0 = np.array([170, 174, 176, 177, 178, 179], np.int32)
    L = synthetic_text(h, B=len(prev0), seed=5)
    Yr, _, trajr = O.synthesize(L, weights, h, np.float32, run_ssrn=False, prev0=prev0)
    assert trajr.max() == h.max_N - 1 and (trajr[:, 0] >= prev0).all() and (np.diff(trajr, axis=1) >= 0).all()
    for graph in (0, 1):
        eng.set_decode_graph(graph)
        eng.debug_seed_prev_max(prev0)
        Y, mx = eng.text2mel(dev(L))
        eng.synchronize()
        np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
        err = maxabs(Y.cpu().numpy(), Yr)
        assert err < TOL, f"mode {mode} graph {graph}: decode max-abs {err}"
    # the seed is consumed: the next decode starts from zeros again
    Y0, mx0 = eng.text2mel(dev(L))
    assert int(mx0[:, 0].max()) <= 2
    eng.set_decode_mode(DEFAULT_MODE)


def test_attention_window_size_is_validated(weights):
    """The decode attention kernels are unrolled for a 3-key window: a larger hp.attention_win_size must be refused at create
    time, not silently truncated (the full Attention() kernel would honour it and disagree with the decode)."""
    from dc_tts_amd.engine import DcttsError, Engine
    with pytest.raises(DcttsError, match="attention_win_size"):
        Engine(weights, hp.replace(attention_win_size=4))


def test_decode_golden_config1(weights):
    """Config 1 of BASELINE.json (Harvard sentence 1) against the committed fixture."""
    g = np.load(os.path.join(GOLD, "config1_harvard1.npz"))
    T = int(g["max_T"])
    eng = engine_for(weights, max_T=T)
    Y, mx = eng.text2mel(dev(g["L"]))
    np.testing.assert_array_equal(mx.cpu().numpy(), g["traj"])
    assert maxabs(Y.cpu().numpy(), g["Y"]) < TOL


def test_reference_loop_driver_equals_fast_path(weights):
    """synthesize.py's literal loop on the drop-in network functions == the one-call incremental decoder."""
    from dc_tts_amd.synthesize import synthesize, synthesize_reference_loop
    T = 90
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=2, seed=33))
    Y1, Z1, t1 = synthesize(L, eng)
    Y2, Z2, t2 = synthesize_reference_loop(L, eng)
    assert torch.equal(t1, t2)
    assert float((Y1 - Y2).abs().max()) < 1e-4 and float((Z1 - Z2).abs().max()) < 1e-3


# ---------------------------------------------------------------- full-size properties (BASELINE configs[1], [3])
def test_full_size_properties(weights):
    """B=32, N=180, T=210 (configs[1]/[3] shard): determinism, shard independence, monotone attention, ranges,
    and oracle agreement on two of the utterances."""
    eng = engine_for(weights)
    eng.set_decode_graph(True)
    Lh = synthetic_text(hp, B=32, seed=1234)
    L = dev(Lh)
    Y, Z, mx = eng.synthesize(L)
    Y2, Z2, mx2 = eng.synthesize(L)
    assert torch.equal(Y, Y2) and torch.equal(Z, Z2) and torch.equal(mx, mx2)          # run-to-run bitwise
    Ya, Za, mxa = eng.synthesize(L[:16].contiguous())
    Yb, Zb, mxb = eng.synthesize(L[16:].contiguous())
    assert torch.equal(torch.cat((Ya, Yb)), Y) and torch.equal(torch.cat((mxa, mxb)), mx)   # shards == whole batch, bitwise
    # SSRN rows are served by the 32-row or the 16-row MFMA kernel depending on where they fall in the launch
    # (hconv16_kernel.h), so across batch sizes Z agrees to fp32 reassociation, not bitwise
    assert float((torch.cat((Za, Zb)) - Z).abs().max()) < 1e-5
    m = mx.cpu().numpy()
    dm = np.diff(m, axis=1)
    assert m.min() >= 0 and m.max() < hp.max_N and dm.min() >= 0 and dm.max() <= hp.attention_win_size - 1
    assert tuple(Z.shape) == (32, 4 * hp.max_T, hp.n_linear)
    z = Z.cpu().numpy()
    assert np.isfinite(z).all() and z.min() >= 0 and z.max() <= 1
    Yr, Zr, trajr = O.synthesize(Lh[:2], weights, hp, np.float32)
    np.testing.assert_array_equal(m[:2], trajr)
    assert maxabs(Y[:2].cpu().numpy(), Yr) < TOL and maxabs(z[:2], Zr) < TOL
    # ... and ALL 32 utterances x 210 frames against the numpy statement of the incremental algorithm (oracle/incremental_ref.py, proven equal to
    # the full-recompute loop by tests/test_oracle.py), with the smallest top-2 logit gap of any arg-max decision of the batch
    from oracle.incremental_ref import incremental_decode_v3
    st = {}
    Yi, traji = incremental_decode_v3(Lh, weights, hp, np.float32, stats=st)
    print(f"headline shape, 32 x 210: min top-2 logit gap {st['min_top2_logit_gap']:.3e}, max|Y - oracle| {maxabs(Y.cpu().numpy(), Yi):.2e}")
    np.testing.assert_array_equal(m, traji)
    assert maxabs(Y.cpu().numpy(), Yi) < TOL
    _, Zi = O.SSRN(Yi[8:12], weights, hp)                         # four more utterances through the SSRN oracle
    assert maxabs(z[8:12], Zi) < TOL


def test_ragged_batch_sizes(weights):
    """Batch sizes that do not fill the 8-row chain tiles / 32-row bulk tiles (B = 3, 33): Text2Mel output and the attention
    trajectory of every utterance are bitwise those of the same utterance decoded in another batch composition."""
    T = 24
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=33, seed=404))
    Y33, m33 = eng.text2mel(L)
    Y32, m32 = eng.text2mel(L[:32].contiguous())
    Y3, m3 = eng.text2mel(L[30:33].contiguous())
    assert torch.equal(Y33[:32], Y32) and torch.equal(m33[:32], m32)
    assert torch.equal(Y33[30:33], Y3) and torch.equal(m33[30:33], m3)
    assert bool(torch.isfinite(Y33).all())


def test_empty_batch_returns_empty_results(weights):
    """An empty feed (B = 0; e.g. a rank of a sharded run that owns no utterance, or a serving loop with nothing queued): the reference's sess.run returns arrays with
    a zero leading dimension and the right trailing shape; so does every function of the drop-in, without a launch.  The C ABI itself keeps rejecting B <= 0."""
    eng = engine_for(weights)
    L0 = torch.zeros((0, hp.max_N), dtype=torch.int32, device="cuda")
    Y, Z, mx = eng.synthesize(L0, check=True)
    assert tuple(Y.shape) == (0, hp.max_T, hp.n_mels) and tuple(Z.shape) == (0, hp.r * hp.max_T, hp.n_linear) and tuple(mx.shape) == (0, hp.max_T) and mx.dtype == torch.int64
    Y2, mx2, al = eng.text2mel(L0, alignments=True)
    assert tuple(Y2.shape) == (0, hp.max_T, hp.n_mels) and tuple(al.shape) == (0, hp.max_N, hp.max_T)
    K, V = eng.text_enc(L0)
    assert tuple(K.shape) == (0, hp.max_N, hp.d) == tuple(V.shape)
    Q = eng.audio_enc(torch.zeros((0, 7, hp.n_mels), device="cuda"))
    R, A, m = eng.attention(Q, torch.zeros((0, 9, hp.d), device="cuda"), torch.zeros((0, 9, hp.d), device="cuda"))
    assert tuple(Q.shape) == (0, 7, hp.d) and tuple(R.shape) == (0, 7, 2 * hp.d) and tuple(A.shape) == (0, 9, 7)
    lg, Yd = eng.audio_dec(R)
    lz, Zs = eng.ssrn(Yd)
    assert tuple(Yd.shape) == (0, 7, hp.n_mels) and tuple(Zs.shape) == (0, 28, hp.n_linear) and tuple(lz.shape) == (0, 28, hp.n_linear)
    eng.synchronize()
    # ... and the context is as usable as before
    L = dev(synthetic_text(hp, B=2, seed=5))
    Ya, _ = eng.text2mel(L); eng.synchronize()
    assert bool(torch.isfinite(Ya).all())
    import ctypes
    assert eng.lib.dctts_text2mel_decode(eng._h, ctypes.c_void_p(L.data_ptr()), 0, hp.max_N, hp.max_T, ctypes.c_void_p(Ya.data_ptr()), None, None, None) < 0


def test_large_batch_team_rounds(weights):
    """B = 134: 34 teams of the decode's team kernels = 640 team workgroups, more than two rounds of the 256 CUs, the last team with two utterances.
    Every utterance's mel rows and trajectory are bitwise those of the same utterance decoded inside a batch of 32 (teams that all fit at once)."""
    T = 100
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=134, seed=77))
    Y, mx = eng.text2mel(L)
    eng.synchronize()
    for a in (0, 64, 102):
        Ys, ms = eng.text2mel(L[a:a + 32].contiguous())
        eng.synchronize()
        assert torch.equal(Y[a:a + 32], Ys) and torch.equal(mx[a:a + 32], ms), f"utterances {a} .. {a + 31}"
    assert bool(torch.isfinite(Y).all())


def test_long_form_shape(weights):
    """configs[4] shape class: max_T = 1000 (one GPU's share, B = 8); cone / history indexing far beyond 210."""
    T = 1000
    eng = engine_for(weights, max_T=T)
    eng.set_decode_graph(True)
    Lh = synthetic_text(hp, B=8, seed=77)
    L = dev(Lh)
    Y, mx = eng.text2mel(L)
    Y2, mx2 = eng.text2mel(L)
    assert torch.equal(Y, Y2) and torch.equal(mx, mx2)
    # BASELINE configs[4] parity over ALL 1000 frames of ALL 8 utterances against the numpy statement of the incremental algorithm
    # (oracle/incremental_ref.py, proven equal to the restated synthesize.py loop on CPU).  With seeded random weights the attention
    # stalls near key 60, so the clipped-window regime is NOT reached here: test_decode_end_of_text_window covers it.
    from oracle.incremental_ref import incremental_decode_v3
    st = {}
    Yr, trajr = incremental_decode_v3(Lh, weights, hp.replace(max_T=T), np.float32, stats=st)      # all 8 utterances x 1000 frames
    print(f"T=1000, 8 x 1000: min top-2 logit gap {st['min_top2_logit_gap']:.3e}, trajectory max {trajr.max()}, max|Y - oracle| {maxabs(Y.cpu().numpy(), Yr):.2e}")
    np.testing.assert_array_equal(mx.cpu().numpy(), trajr)
    assert maxabs(Y.cpu().numpy(), Yr) < TOL
    m = mx.cpu().numpy()
    assert (np.diff(m, axis=1) >= 0).all() and m.max() < hp.max_N
    y = Y.cpu().numpy()
    assert np.isfinite(y).all() and y.min() >= 0 and y.max() <= 1
    # prefix property of a causal autoregressive decoder: the first 210 frames equal the T=210 run
    e2 = engine_for(weights)
    Ys, _ = e2.text2mel(L)
    assert torch.equal(Ys, Y[:, :hp.max_T])
    # ... which also ties this run to the oracle's FULL-RECOMPUTE loop (the restated synthesize.py loop itself, not the numpy model of the
    # incremental algorithm): the first 260 frames of utterance 0 against O.synthesize at max_T = 260 (> 210: past the bench shape, and past
    # the 173-frame receptive field of AudioEnc plus the 85-frame cone of AudioDec)
    Tp = 260
    Yo, _, trajo = O.synthesize(Lh[:1], weights, hp.replace(max_T=Tp), np.float32, run_ssrn=False)
    np.testing.assert_array_equal(mx[:1, :Tp].cpu().numpy(), trajo)
    assert maxabs(Y[:1, :Tp].cpu().numpy(), Yo) < TOL
    # SSRN at the long-form shape (networks.py:214-292 at configs[4]: (8, 1000, 80) -> (8, 4000, 1025)): the whole batch through the HIP path, one utterance's
    # (4000, 1025) against the oracle -- the synthesize() call of this shape, Z included, not only the decode
    Y3, Z3, m3 = eng.synthesize(L)
    eng.synchronize()
    assert torch.equal(Y3, Y) and torch.equal(m3, mx) and tuple(Z3.shape) == (8, 4 * T, hp.n_linear)
    u = 5
    _, Zr = O.SSRN(Y3[u:u + 1].cpu().numpy(), weights, hp.replace(max_T=T), np.float32)
    ez = maxabs(Z3[u:u + 1].cpu().numpy(), Zr)
    print(f"T=1000: max|Z - oracle| of utterance {u} over (4000, 1025): {ez:.2e}")
    assert ez < TOL and bool(torch.isfinite(Z3).all())


# ---------------------------------------------------------------- multi-rank plumbing on one GPU (the driver runs the 8-GPU scaling)
_SHARD_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.sharding import gpu_synth, synthesize_sharded
from dc_tts_amd.weights import synthetic_text, synthetic_weights
dist.init_process_group("gloo")
torch.cuda.set_device(0)                                     # both ranks share the one GPU of the test box
h = hp.replace(max_T=20)
eng = Engine(synthetic_weights(h, seed=1234, perturb=True), h, device=0)
eng.set_team_kernels(False)                                  # two PROCESSES on one GPU: their team kernels would wait for each other's compute units (one launch per layer instead)
L = synthetic_text(h, B=5, seed=42)                          # ragged: ranks get 3 and 2 utterances
out = synthesize_sharded(L, gpu_synth(eng))
if dist.get_rank() == 0:
    np.savez(sys.argv[2], Y=out[0], Z=out[1], traj=out[2])
    print("SHARD_GPU_OK")
else:
    assert out is None
dist.destroy_process_group()
'''


def test_two_ranks_share_gpu_gather_equals_single_process(weights, tmp_path):
    """SURVEY 8e end to end on one GPU: two ranks (gloo, both on cuda:0) decode their contiguous slices, copy the results into
    pinned host buffers and meet on rank 0's host; (Y, trajectory) must be bitwise the single-process result, Z to the 1e-5 of
    the SSRN row-split (hconv16_kernel.h).  Both sides decode with one launch per layer: team kernels of two processes on ONE GPU hold compute units
    the other process's teams wait for (the driver's ranks have a GPU each; inside a process the device lease serialises decodes), and a decode repeated in
    the per-layer form is equal to the team form to 3e-6, not bitwise."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"; script.write_text(_SHARD_WORKER)
    outp = str(tmp_path / "out.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29621", str(script), root, outp]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SHARD_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    g = np.load(outp)
    T = 20
    eng = engine_for(weights, max_T=T)
    L = synthetic_text(hp.replace(max_T=T), B=5, seed=42)
    eng.set_team_kernels(False)
    try:
        Y, Z, mx = eng.synthesize(dev(L))
        eng.synchronize()
    finally:
        eng.set_team_kernels(True)
    np.testing.assert_array_equal(g["traj"], mx.cpu().numpy())
    np.testing.assert_array_equal(g["Y"], Y.cpu().numpy())
    assert maxabs(g["Z"], Z.cpu().numpy()) < 1e-5


def test_bench_two_ranks_through_torch_distributed_run():
    """The driver's multi-GPU launch line on the one GPU of the test box: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
    (gloo, both ranks on cuda:0).  Rank 0's untimed extras run while rank 1 sits in dist.barrier(): nothing may time out; ONE JSON line with
    n_gpus = 2, weak scaling, the whole-job value, the gather, the roofline object."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--max-T", "24", "--no-cpu-baseline", "--no-vocoder", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 2 and out["dtype"] == "f32"
    assert abs(out["value"] - 2 * 8 * 24 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-3         # whole-job frames over the max-over-ranks time
    assert out["gather"]["bytes_per_rank"] == 8 * 96 * 1025 * 4 and "roofline" in out and out["config"]["sharding"].startswith("2 x 8")


# ---------------------------------------------------------------- the reference's literal workload, from the reference's own run
def test_harvard20_against_the_reference_run(weights):
    """synthesize.py:23: all 20 Harvard sentences as ONE batch at the shipped hyper-parameters (N = 180, T = 210).  The fixture
    (tests/golden/harvard20_ref.npz) was produced by executing the reference's synthesize.synthesize() itself on the numpy TensorFlow
    stand-in (tests/golden/make_golden_from_reference.py): L from its load_data, Y and the trajectory from its loop, Z from its SSRN pass,
    `alignments` as its last loop step fetched them."""
    g = np.load(os.path.join(GOLD, "harvard20_ref.npz"))
    eng = engine_for(weights)
    eng.set_decode_graph(False)
    L = dev(g["L"])
    assert tuple(L.shape) == (20, hp.max_N)
    Y, Z, mx, al = eng.synthesize(L, alignments=True, check=True)
    np.testing.assert_array_equal(mx.cpu().numpy(), g["traj"])
    ey = maxabs(Y.cpu().numpy(), g["Y"])
    sy, sx = (int(v) for v in g["z_sub_step"])
    ez = maxabs(Z.cpu().numpy()[:, ::sy, ::sx], g["Z_sub"])
    ea = maxabs(al.cpu().numpy(), g["alignments_last"])
    print(f"harvard20: max|Y - ref| {ey:.2e}, max|Z - ref| {ez:.2e}, max|alignments - ref| {ea:.2e}, trajectory ends at {g['traj'][:, -1].tolist()}")
    assert ey < TOL and ez < TOL and ea < 1e-4
    a = al.cpu().numpy()
    assert a.shape == (20, hp.max_N, hp.max_T) and np.allclose(a.sum(1), 1.0, atol=1e-5)
    # the same batch through the text front-end of the product (data_load.py:79-86) when the sentences file is at hand
    # (it is reference DATA, read only where /root/reference exists; the fixture carries L for the GPU box)


def test_alignments_output_equals_the_boundary_attention(weights):
    """`alignments` of the decode == Attention(Q, K, V, True, prev) of the boundary function on the decode's own Q history with the
    window of the last step (networks.py:142-153): checked through the literal loop driver, whose last step computes exactly that."""
    from dc_tts_amd import networks
    T = 40
    eng = engine_for(weights, max_T=T)
    networks.bind(eng)
    L = dev(synthetic_text(hp.replace(max_T=T), B=3, seed=5))
    Y, mx, al = eng.text2mel(L, alignments=True, check=True)
    K, V = networks.TextEnc(L, training=False)
    S = torch.cat((torch.zeros_like(Y[:, :1]), Y[:, :-1]), 1).contiguous()          # train.py:51
    Q = networks.AudioEnc(S, training=False)
    prev = mx[:, T - 2].to(torch.int32).contiguous()                                # the value fed at the last step
    _, al_ref, mx_ref = networks.Attention(Q, K, V, mononotic_attention=True, prev_max_attentions=prev)
    assert float((al - al_ref).abs().max()) < 1e-5
    assert torch.equal(mx_ref[:, T - 1], mx[:, T - 1])


# ---------------------------------------------------------------- a failed decode cannot be consumed silently
def test_failed_decode_is_poisoned_reported_and_recovered(weights):
    """dctts_debug_inject_decode_error makes the next decode end as a real in-launch time-out does: every output is NaN / -1, the sticky
    status reports it (once), and the decode after that is clean and bitwise equal to an undisturbed one."""
    from dc_tts_amd.engine import DcttsError
    T = 30
    eng = engine_for(weights, max_T=T)
    eng.set_decode_graph(False)
    L = dev(synthetic_text(hp.replace(max_T=T), B=5, seed=8))
    Y0, Z0, m0, a0 = eng.synthesize(L, alignments=True)
    eng.synchronize()
    eng.debug_inject_decode_error(1)
    Y1, Z1, m1, a1 = eng.synthesize(L, alignments=True)
    torch.cuda.synchronize()
    assert bool(torch.isnan(Y1).all()) and bool(torch.isnan(Z1).all()) and bool(torch.isnan(a1).all()) and bool((m1 == -1).all())
    with pytest.raises(DcttsError, match="failed on the device"):
        eng.synchronize()
    eng.synchronize()                                            # reported once, then clean
    Y2, Z2, m2, a2 = eng.synthesize(L, alignments=True)
    eng.synchronize()
    assert torch.equal(Y2, Y0) and torch.equal(Z2, Z0) and torch.equal(m2, m0) and torch.equal(a2, a0)


def test_a_failure_survives_back_to_back_decodes(weights):
    """The advisor's case: decode k fails, decode k + 1 is enqueued before anybody synchronises.  The failure of k must still be reported
    (the status word is sticky on the device and never cleared by a decode), k's outputs are poisoned, k + 1's are valid."""
    from dc_tts_amd.engine import DcttsError
    T = 30
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=4, seed=9))
    Yg, mg = eng.text2mel(L)
    eng.synchronize()
    eng.debug_inject_decode_error(4)
    Ya, ma = eng.text2mel(L)                                     # fails (injected)
    Yb, mb = eng.text2mel(L)                                     # enqueued behind it without a host synchronisation
    with pytest.raises(DcttsError, match="1 decode"):
        eng.synchronize()
    assert bool(torch.isnan(Ya).all()) and bool((ma == -1).all())
    assert torch.equal(Yb, Yg) and torch.equal(mb, mg)
    # ... and a failure nobody asks about refuses the next decode call once
    eng.debug_inject_decode_error(1)
    eng.text2mel(L)
    torch.cuda.synchronize()
    with pytest.raises(DcttsError, match="EARLIER decode"):
        eng.text2mel(L)
    Yc, mc = eng.text2mel(L)
    eng.synchronize()
    assert torch.equal(Yc, Yg) and torch.equal(mc, mg)


def test_checked_decode_repeats_a_failed_decode_once(weights):
    """Engine.text2mel(check=True) = the reference's synchronous sess.run: a decode that failed on the device is repeated with one launch per layer
    (the form without in-launch hand-offs) and the caller gets valid results; the team kernels are back on afterwards."""
    T = 30
    eng = engine_for(weights, max_T=T)
    Lh = synthetic_text(hp.replace(max_T=T), B=4, seed=10)
    L = dev(Lh)
    Yg, mg = eng.text2mel(L)
    eng.synchronize()
    eng.debug_inject_decode_error(16)
    Y, mx = eng.text2mel(L, check=True)
    assert bool(torch.isfinite(Y).all()) and torch.equal(mx, mg) and float((Y - Yg).abs().max()) < 1e-5
    Yr, traj, _ = oracle_loop_with_margin(Lh, weights, hp.replace(max_T=T))
    np.testing.assert_array_equal(mx.cpu().numpy(), traj)
    assert maxabs(Y.cpu().numpy(), Yr) < TOL
    Y3, m3 = eng.text2mel(L)
    eng.synchronize()
    assert torch.equal(Y3, Yg) and torch.equal(m3, mg)


def test_two_contexts_on_one_gpu_take_turns(weights):
    """Two engines in one process decode 'at the same time' on two torch streams: the library serialises the decodes of different contexts on a
    device (their team kernels would starve each other), so both finish valid and bitwise equal to solo runs."""
    from dc_tts_amd.engine import Engine
    T = 40
    h = hp.replace(max_T=T)
    e1 = engine_for(weights, max_T=T)
    e2 = Engine(weights, h)
    L1 = dev(synthetic_text(h, B=32, seed=21)); L2 = dev(synthetic_text(h, B=32, seed=22))
    Y1s, m1s = e1.text2mel(L1); e1.synchronize()
    Y2s, m2s = e2.text2mel(L2); e2.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            outs.append(("a",) + tuple(e1.text2mel(L1)))
        with torch.cuda.stream(s2):
            outs.append(("b",) + tuple(e2.text2mel(L2)))
    torch.cuda.synchronize()
    e1.decode_status(); e2.decode_status()
    for tag, Y, m in outs:
        Ys, ms = (Y1s, m1s) if tag == "a" else (Y2s, m2s)
        assert torch.equal(Y, Ys) and torch.equal(m, ms)
    e2.close()


def test_decode_from_a_high_priority_stream_and_beside_ssrn_on_another_stream(weights):
    """Stream handling of the decode (streams and priorities): called from a default-priority stream the chain's launches run on a high-priority
    stream of the context between two events, called from a high-priority stream they run on that stream itself -- same result, bitwise.  And one context serves
    two streams at once: SSRN of the previous batch on a second stream beside TextEnc + decode of the next one (their scratch buffers are separate: with one shared
    buffer for SSRN's tap-split tails and TextEnc's column split this test -- and 18 % of tools/soak.py's phase C -- returned different values, unreported)."""
    T = 60
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=32, seed=31))
    Y, mx = eng.text2mel(L)
    Z = eng.ssrn(Y, want_logits=False)[1]
    eng.synchronize()
    hi = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(hi):
        Yh, mh = eng.text2mel(L)
    torch.cuda.synchronize()
    assert torch.equal(Yh, Y) and torch.equal(mh, mx)
    s2 = torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(s2):
            Zs = [eng.ssrn(Y, want_logits=False)[1] for _ in range(3)]
        Y2, m2 = eng.text2mel(L)
        torch.cuda.synchronize()
        eng.decode_status()
        assert torch.equal(Y2, Y) and torch.equal(m2, mx)
        for z in Zs: assert torch.equal(z, Z)


def test_new_engines_get_two_streams_that_really_run_concurrently(weights):
    """Round 6: the decode's two streams wait for each other, so they must sit on different hardware queues -- and HIP lets a NEW high-priority stream share the queue
    of an existing one once its pool of queues is full.  With one engine's pair and a caller's own high-priority stream alive, the next engines' pairs landed on ONE
    queue: every decode of theirs ran into its bounded waits (error word 36; rounds 1-5 had the same hole, tools/ctx_probe.py).  The pair is now tested when it is
    created (a kernel on one stream waits for a flag set by a kernel on the other) and the side stream replaced until it passes; a caller's high-priority stream is
    tested against the side stream the same way before the chain is put on it.  Here: exactly that history, then several new engines, each decoding at once -- in
    the default form, with stream-operation meetings (which would HANG on a shared queue) and in the per-layer fallback form."""
    from dc_tts_amd.engine import Engine
    T = 40
    h = hp.replace(max_T=T)
    e1 = engine_for(weights, max_T=T)
    L = dev(synthetic_text(h, B=32, seed=31))
    Yg, mg = e1.text2mel(L); e1.synchronize()
    his = [torch.cuda.Stream(priority=-1) for _ in range(3)]        # (kept alive: they hold hardware queues of the high-priority pool)
    for s in his:
        with torch.cuda.stream(s):
            Yh, mh = e1.text2mel(L)
        torch.cuda.synchronize(); e1.decode_status()
        assert torch.equal(Yh, Yg) and torch.equal(mh, mg)
    for k, knob in enumerate(["", "", "DCTTS_CHAIN_WAIT=0", "DCTTS_XGROUP=0,DCTTS_XCONE=0", "", ""]):
        pairs = [kv.split("=") for kv in knob.split(",")] if knob else []
        for name, val in pairs: os.environ[name] = val
        try:
            e2 = Engine(weights, h)
        finally:
            for name, _ in pairs: del os.environ[name]
        Y2, m2 = e2.text2mel(L); e2.synchronize()
        assert torch.equal(m2, mg) and maxabs(Y2.cpu().numpy(), Yg.cpu().numpy()) < 1e-5, (k, knob)
        with torch.cuda.stream(his[k % 3]):                          # ... and from a caller's high-priority stream
            Y3, m3 = e2.text2mel(L)
        torch.cuda.synchronize(); e2.decode_status()
        assert torch.equal(m3, mg) and maxabs(Y3.cpu().numpy(), Yg.cpu().numpy()) < 1e-5, (k, knob)
        e2.close()


def test_checked_retry_leaves_the_team_kernel_settings_alone(weights):
    """the safe retry of a checked decode is a ONE-SHOT form of that decode (dctts_decode_safe_once); it must not switch the team kernels back on for a
    caller who switched them off, nor re-enable them after dctts_decode_status switched them off for good."""
    T = 24
    eng = engine_for(weights, max_T=T)
    L = dev(synthetic_text(hp.replace(max_T=T), B=4, seed=12))
    Yg, mg = eng.text2mel(L)
    eng.synchronize()
    assert eng.debug_team_kernels_state() == 7
    eng.set_team_kernels(False)
    try:
        assert eng.debug_team_kernels_state() & 3 == 0
        eng.debug_inject_decode_error(16)
        Y, mx = eng.text2mel(L, check=True)
        assert eng.debug_team_kernels_state() & 3 == 0, "the retry switched the team kernels back on"
        assert torch.equal(mx, mg) and float((Y - Yg).abs().max()) < 1e-5
    finally:
        eng.set_team_kernels(True)
    assert eng.debug_team_kernels_state() == 7
    eng.debug_inject_decode_error(16)
    Y, mx = eng.text2mel(L, check=True)                       # with them on: retried in the safe form, still on afterwards
    assert eng.debug_team_kernels_state() == 7 and torch.equal(mx, mg) and float((Y - Yg).abs().max()) < 1e-5
    Y3, m3 = eng.text2mel(L)
    eng.synchronize()
    assert torch.equal(Y3, Yg) and torch.equal(m3, mg)


def test_one_engine_two_streams_same_kind_calls_of_different_inputs(weights):
    """2 x SSRN, 2 x TextEnc and 2 x synthesize of DIFFERENT inputs enqueued concurrently on two streams of ONE engine are bitwise equal to solo runs.
    Calls that share scratch memory are ordered on the device by the context (use groups, include/dctts_hip.h "streams and threads"), whatever streams they come from."""
    T = 40
    h = hp.replace(max_T=T)
    eng = engine_for(weights, max_T=T)
    La, Lb = dev(synthetic_text(h, B=32, seed=41)), dev(synthetic_text(h, B=32, seed=42))
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    Ma, Mb = torch.rand(32, T, hp.n_mels, generator=g).cuda(), torch.rand(32, T, hp.n_mels, generator=g).cuda()
    solo = {}
    for tag, L, M in (("a", La, Ma), ("b", Lb, Mb)):
        solo[tag] = (eng.text_enc(L), eng.ssrn(M, want_logits=False)[1], eng.synthesize(L))
    eng.synchronize()
    assert not torch.equal(solo["a"][1], solo["b"][1]) and not torch.equal(solo["a"][2][0], solo["b"][2][0])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    torch.cuda.synchronize()
    outs = []
    for rnd in range(3):
        order = ((s1, "a", La, Ma), (s2, "b", Lb, Mb)) if rnd % 2 == 0 else ((s2, "b", Lb, Mb), (s1, "a", La, Ma))
        for what in ("ssrn", "te", "syn"):                     # interleaved: a's SSRN, b's SSRN, a's TextEnc, b's TextEnc, a's synthesize, b's synthesize
            for st, tag, L, M in order:
                with torch.cuda.stream(st):
                    if what == "ssrn": outs.append((tag, 1, eng.ssrn(M, want_logits=False)[1]))
                    elif what == "te": outs.append((tag, 0, eng.text_enc(L)))
                    else: outs.append((tag, 2, eng.synthesize(L)))
    torch.cuda.synchronize()
    eng.decode_status()
    for tag, k, got in outs:
        ref = solo[tag][k]
        if k == 1: assert torch.equal(got, ref)
        else: assert all(torch.equal(x, y) for x, y in zip(got, ref)), (tag, k)


def test_first_decode_at_a_new_geometry_is_valid(weights):
    """Found by bench.py in round 5: the first decode at a batch size not seen before allocated the team kernels' exchange memory and zeroed it with hipMemset, which may
    run AFTER the call returns (null stream) -- i.e. in the middle of the decode, wiping its team barriers (error word 1).  Rounds 3-4 hid that behind the
    hipDeviceSynchronize of every table rebuild.  Every geometry here is new to a fresh engine; each first decode must be valid and equal to its repetition."""
    from dc_tts_amd.engine import Engine
    eng = Engine(weights, hp.replace(max_T=64))
    for B, T in ((128, 40), (70, 33), (5, 64), (37, 21), (64, 48), (96, 17)):
        L = dev(synthetic_text(hp.replace(max_T=T), B=B, seed=B + T))
        Y1, m1 = eng.text2mel(L, max_T=T)
        eng.synchronize()                                            # raises if the decode failed on the device
        Y2, m2 = eng.text2mel(L, max_T=T)
        eng.synchronize()
        assert bool(torch.isfinite(Y1).all()) and torch.equal(Y1, Y2) and torch.equal(m1, m2), (B, T)
    eng.close()


def test_alternating_batch_shapes_reuse_their_workspaces(weights):
    """Workspaces and the decode's tables are cached per geometry: B = 32 / B = 6 / T changes alternate without the cache growing after the first round, results
    bitwise equal each time; past dctts_set_workspace_limit the cache is dropped (one device sync) and everything still works."""
    eng = engine_for(weights, max_T=30)
    h = hp.replace(max_T=30)
    L32, L6 = dev(synthetic_text(h, B=32, seed=51)), dev(synthetic_text(h, B=6, seed=52))
    first = None
    sizes = []
    for rnd in range(3):
        res = (eng.synthesize(L32), eng.synthesize(L6), eng.synthesize(L32, max_T=18), eng.text2mel(L6, max_T=18))
        eng.synchronize()
        sizes.append(eng.device_bytes())
        if first is None: first = res
        else:
            for a, b in zip(first, res): assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert sizes[1] == sizes[0] and sizes[2] == sizes[0], sizes
    eng.set_workspace_limit(1 << 20)                            # everything cached is over 1 MiB: the next call starts from an empty cache
    try:
        res = (eng.synthesize(L32), eng.synthesize(L6))
        eng.synchronize()
        for a, b in zip(first[:2], res): assert all(torch.equal(x, y) for x, y in zip(a, b))
        assert eng.device_bytes() < sizes[0]
    finally:
        eng.set_workspace_limit(96 << 30)


# ---------------------------------------------------------------- the opt-in split-bf16 contraction (dctts_set_split_bf16)
_bf_engine = {}


def bf16_engine(weights):
    from dc_tts_amd.engine import Engine
    if "e" not in _bf_engine:
        _bf_engine["e"] = Engine(weights, hp, split_bf16=2)
    return _bf_engine["e"]


BF_CASES = [c for c in CASES if c.values[0] in ("textenc", "ssrn")]


@pytest.mark.parametrize("net,scope,causal,li", BF_CASES)
def test_split_bf16_layer_vs_oracle(weights, net, scope, causal, li):
    """Every layer shape class of TextEnc and SSRN on the split-bf16 form (hi.hi + hi.mid + mid.hi on the bf16 matrix pipe, fp32 accumulate) against the oracle at the
    SAME tolerance as the fp32 form (2e-4), whole 32-row items incl. a ragged one; the fp32 form of the same engine must still be selectable and exact."""
    fn = {"textenc": textenc_layers, "ssrn": ssrn_layers}[net]
    layers = fn(hp)
    l = layers[li]
    eng = bf16_engine(weights)
    rng = np.random.default_rng(100 + li)
    B, T = 2, 75
    P = O._Scoped(weights, scope, np.float32)
    di = _dev_index(layers, li, net)
    if net == "textenc" and li == 1:
        ids = rng.integers(0, len(hp.vocab), (B, T)).astype(np.int32)
        ref = O.conv1d(O.embed(ids, P["embed_1/lookup_table"]), P, "C_2", act=O.relu)
        run = lambda: eng.debug_layer(net, 0, dev(ids), l.cout).cpu().numpy()
    else:
        x = rng.standard_normal((B, T, l.cin)).astype(np.float32)
        if l.kind == "C":
            ref = O.conv1d(x, P, l.scope, padding="SAME", act=O.relu if l.act == "relu" else None)
            run = lambda: eng.debug_layer(net, di, dev(x), l.cout).cpu().numpy()
        elif l.kind == "HC":
            ref = O.hc(x, P, l.scope, rate=l.rate, padding="SAME")
            run = lambda: eng.debug_layer(net, di, dev(x), l.cout).cpu().numpy()
        else:
            ref = O.conv1d_transpose(x, P, l.scope)
            run = lambda: eng.debug_layer(net, di, dev(x), l.cout, upsample=2).cpu().numpy()
    if net == "ssrn" and li == len(layers) - 1:
        ref = O.sigmoid(ref)
    eng.set_split_bf16(2)
    got = run()
    eng.set_split_bf16(0)
    got32 = run()
    eng.set_split_bf16(2)
    e, e32 = maxabs(got, ref), maxabs(got32, ref)
    print(f"{net}/{l.scope}: split-bf16 max-abs {e:.2e} (fp32 form {e32:.2e})")
    assert got.shape == ref.shape and e < 2e-4 and e32 < 2e-4, (e, e32)
    assert not np.array_equal(got, got32) or l.cin < 16       # (the two forms are different arithmetic: equal outputs would mean the switch does nothing)


def test_split_bf16_pipeline_error_and_untouched_decode(weights):
    """Mode 1 (SSRN only): Text2Mel -- mel frames AND attention trajectory -- is bit-identical to the exact form, Z within the north-star 1e-3 of the float64 oracle;
    mode 2 (+ TextEnc): the trajectory is still the oracle's, Y and Z within 1e-3.  The measured errors are printed (DESIGN.md section 11 quotes them)."""
    T = 50
    h = hp.replace(max_T=T)
    from dc_tts_amd.engine import Engine
    eng = Engine(weights, h, split_bf16=2)
    Lh = synthetic_text(h, B=4, seed=77)
    L = dev(Lh)
    eng.set_split_bf16(0); Y0, Z0, m0 = eng.synthesize(L); eng.synchronize()
    eng.set_split_bf16(1); Y1, Z1, m1 = eng.synthesize(L); eng.synchronize()
    eng.set_split_bf16(2); Y2, Z2, m2 = eng.synthesize(L); eng.synchronize()
    assert torch.equal(Y1, Y0) and torch.equal(m1, m0) and not torch.equal(Z1, Z0)
    W64 = {k: v.astype(np.float64) for k, v in weights.items()}
    Yr, Zr, traj = O.synthesize(Lh, W64, h, np.float64)
    np.testing.assert_array_equal(m0.cpu().numpy(), traj); np.testing.assert_array_equal(m2.cpu().numpy(), traj)
    errs = {"fp32": (maxabs(Y0.cpu().numpy(), Yr), maxabs(Z0.cpu().numpy(), Zr)), "ssrn": (maxabs(Y1.cpu().numpy(), Yr), maxabs(Z1.cpu().numpy(), Zr)),
            "ssrn+textenc": (maxabs(Y2.cpu().numpy(), Yr), maxabs(Z2.cpu().numpy(), Zr))}
    print("max-abs vs the float64 oracle (Y, Z):", {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in errs.items()})
    for k, (ey, ez) in errs.items():
        assert ey < TOL and ez < TOL, (k, ey, ez)
    eng.close()


def test_split_bf16_needs_the_packing(weights):
    eng = engine_for(weights)                                     # created without split_bf16: no bf16 packing on the device
    with pytest.raises(Exception):
        eng.set_split_bf16(1)
    eng.set_split_bf16(0)


# ---------------------------------------------------------------- the restore path end to end (synthesize.py:32-40)
def test_checkpoint_directories_to_spectrograms(weights, tmp_path):
    """`python -m dc_tts_amd.synthesize --logdir <prefix>`: two checkpoint directories laid out like hp.logdir-1 / hp.logdir-2 -- written by the
    INDEPENDENT bundle encoder of tests/test_tf_checkpoint.py (two shards each, shortened index keys, optimizer slots and gs/global_step beside the
    variables) -> tf.train.latest_checkpoint -> the two name-restricted restores -> Engine -> decode + SSRN; the spectrograms on disk are bitwise
    those of an Engine fed the weight dict directly."""
    import importlib.util
    from dc_tts_amd import tf_checkpoint as C
    from dc_tts_amd.synthesize import main as synth_main
    spec = importlib.util.spec_from_file_location("ttc", os.path.join(os.path.dirname(__file__), "test_tf_checkpoint.py"))
    ttc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ttc)
    logdir = str(tmp_path / "logdir" / "LJ01")
    rng = np.random.default_rng(0)
    for suffix, scope, step in (("-1", "Text2Mel/", 800), ("-2", "SSRN/", 310)):
        d = logdir + suffix
        os.makedirs(d)
        t = {k: v for k, v in weights.items() if k.startswith(scope)}
        for k in list(t)[:40]:                                            # Adam slots of some variables: never restored (TRAINABLE_VARIABLES only)
            t[k + "/Adam"] = rng.standard_normal(t[k].shape).astype(np.float32)
            t[k + "/Adam_1"] = rng.standard_normal(t[k].shape).astype(np.float32)
        t["gs/global_step"] = np.asarray(step * 1000, np.int32)
        t["beta1_power"] = np.asarray(0.5, np.float32)
        ttc.write_bundle(os.path.join(d, f"model_gs_{step}k"), t, keys_per_block=7, num_shards=2, restart_interval=16, tensor_crc=C.crc32c)
        ttc.write_bundle(os.path.join(d, "model_gs_001k"), {k: np.zeros_like(v) for k, v in t.items()}, keys_per_block=50, with_crc=False)   # an older checkpoint that must NOT be picked
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write(f'model_checkpoint_path: "model_gs_{step}k"\nall_model_checkpoint_paths: "model_gs_001k"\nall_model_checkpoint_paths: "model_gs_{step}k"\n')
    text = tmp_path / "sents.txt"
    text.write_text("header\n1. The birch canoe slid on the smooth planks.\n2. Glue the sheet to the dark blue background.\n3. It's easy to tell the depth of a well.\n", encoding="utf-8")
    out = str(tmp_path / "samples")
    synth_main(["--text", str(text), "--logdir", logdir, "--out", out, "--no-wav"])
    from dc_tts_amd.data_load import load_data
    L = load_data("synthesize", str(text), hp)
    eng = engine_for(weights)
    Y, Z, _ = eng.synthesize(dev(L), check=True)
    for i in range(3):
        assert np.array_equal(np.load(os.path.join(out, f"{i + 1}.mel.npy")), Y[i].cpu().numpy())
        assert np.array_equal(np.load(os.path.join(out, f"{i + 1}.mag.npy")), Z[i].cpu().numpy())
    g = np.load(os.path.join(GOLD, "harvard20_ref.npz"))                  # ... and these three sentences are rows 0-2 of the reference's own run
    assert np.array_equal(L, g["L"][:3])
    assert maxabs(Y.cpu().numpy(), g["Y"][:3]) < TOL
