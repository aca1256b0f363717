"""Time of the first training slice at the shapes training would run it at (B = 32, T = 210, train.py / hyperparams.py:45-46):
hc backward for the three highway-block widths of the model, the Text2Mel losses, one Adam step; HIP-event timed."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.train import TrainOps
ops = TrainOps()
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("| call | shape | ms | GFLOP | TFLOP/s | of the fp32 MFMA peak |\n|---|---|---|---|---|---|")
for (B, T, C, k, rate, pad, what) in [(32, 210, 256, 3, 3, "causal", "AudioEnc / AudioDec HC"), (32, 180, 512, 3, 3, "same", "TextEnc HC"),
                                      (32, 210, 512, 3, 1, "same", "SSRN HC (T rows)"), (32, 840, 1024, 3, 1, "same", "SSRN HC_11/12 (4T rows)")]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, T, C, device="cuda", generator=g); dy = torch.randn(B, T, C, device="cuda", generator=g)
    p = {"kernel": torch.randn(k, C, 2 * C, device="cuda", generator=g) * (k * C) ** -0.5, "bias": torch.zeros(2 * C, device="cuda"),
         "g1": torch.ones(C, device="cuda"), "b1": torch.zeros(C, device="cuda"), "g2": torch.ones(C, device="cuda"), "b2": torch.zeros(C, device="cuda")}
    ms = timed(lambda: ops.hc_backward(x, dy, p, rate=rate, padding=pad))
    flop = 3 * 2.0 * B * T * k * C * 2 * C            # forward recompute + dgrad + wgrad
    print(f"| hc_backward ({what}) | B={B} T={T} C={C} k={k} | {ms:.3f} | {flop / 1e9:.1f} | {flop / ms / 1e9:.1f} | {flop / ms / 1e9 / 157.3:.3f} |")
B, T, M, N = 32, 210, 80, 180
lg = torch.randn(B, T, M, device="cuda"); Y = torch.sigmoid(lg); mels = torch.rand(B, T, M, device="cuda"); al = torch.softmax(torch.randn(B, N, T, device="cuda"), 1)
ms = timed(lambda: ops.text2mel_losses(Y, lg, mels, al, 180, 210))
print(f"| text2mel_losses | B={B} T={T} n_mels={M} N={N} | {ms:.3f} | | | ({(5 * Y.numel() + 2 * al.numel()) * 4 / ms / 1e6:.0f} GB/s of algorithmic bytes) |")
n = 52_380_671
var = torch.zeros(n, device="cuda"); grad = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
ms = timed(lambda: ops.adam_step(var, grad, m, v, 1, 1e-3))
print(f"| adam_step (all {n} parameters as one array) | | {ms:.3f} | | | ({7 * n * 4 / ms / 1e6:.0f} GB/s = {7 * n * 4 / ms / 1e6 / 8000:.2f} of the HBM roof) |")

# ---- whole-network reverse passes at training shapes (B = 32, T = 210, N = 180): activations are random tensors of the right shapes
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.layers import audiodec_layers, audioenc_layers, ssrn_layers, textenc_layers
from dc_tts_amd.train import network_backward
from dc_tts_amd.weights import synthetic_weights
Wd = {n: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for n, v in synthetic_weights(hp, seed=1).items()}
def acts(layers, B, T, x0):
    xs, t = [], T
    for L in layers:
        xs.append(x0 if L.kind == "E" else torch.randn(B, t, L.cin, device="cuda"))
        if L.kind == "D": t *= 2
    return xs, t
B, T, N = 32, 210, 180
ids = torch.randint(1, len(hp.vocab), (B, N), dtype=torch.int32, device="cuda")
nets = [("SSRN", ssrn_layers(hp), "SSRN", T, None, "same"), ("AudioDec", audiodec_layers(hp), "Text2Mel/AudioDec", T, None, "causal"),
        ("AudioEnc", audioenc_layers(hp), "Text2Mel/AudioEnc", T, None, "causal"), ("TextEnc", textenc_layers(hp), "Text2Mel/TextEnc", N, ids, "same")]
total = {}
for name, layers, prefix, rows, x0, pad in nets:
    xs, t_out = acts(layers, B, rows, x0)
    dy = torch.randn(B, t_out, layers[-1].cout, device="cuda")
    ms = timed(lambda: network_backward(ops, layers, Wd, prefix, xs, dy, pad), n=3)
    flop = sum(3 * 2.0 * B * (rows * (2 if False else 1)) * L.size * L.cin * L.conv_filters for L in layers if L.kind in ("C", "HC"))
    total[name] = ms
    print(f"| reverse pass over {name} ({len(layers)} layers) | B={B} rows={rows} | {ms:.2f} | | | |")
Q = torch.randn(B, T, hp.d, device="cuda"); K = torch.randn(B, N, hp.d, device="cuda"); V = torch.randn(B, N, hp.d, device="cuda")
dR = torch.randn(B, T, 2 * hp.d, device="cuda"); dAl = torch.randn(B, N, T, device="cuda")
ms = timed(lambda: ops.attention_backward(Q, K, V, dR, dAl))
print(f"| attention_backward | B={B} T={T} N={N} d={hp.d} | {ms:.3f} | | | |")
print(f"| **Text2Mel gradients** (AudioDec + attention + AudioEnc + TextEnc + losses) | | {total['AudioDec'] + total['AudioEnc'] + total['TextEnc'] + ms + 0.03:.2f} | | | |")
print(f"| **SSRN gradients** (reverse pass + losses) | | {total['SSRN'] + 0.1:.2f} | | | |")

# ---- one whole training step (train.py: sess.run(g.train_op)) at the training batch
from dc_tts_amd.train import TrainGraph
Wn = synthetic_weights(hp, seed=1)
for num, name in ((1, "Text2Mel"), (2, "SSRN")):
    g = TrainGraph(num, Wn, hp)
    if num == 1:
        batch = (ids, torch.rand(B, T, hp.n_mels, device="cuda"))
    else:
        batch = (torch.rand(B, T, hp.n_mels, device="cuda"), torch.rand(B, 4 * T, hp.n_linear, device="cuda"))
    ms = timed(lambda: g.train_op(*batch), n=3)
    print(f"| **one {name} training step** (forward keeping activations, losses, all gradients, clip + Adam) | B={B} T={T} | {ms:.1f} | | | {B / (ms * 1e-3):.0f} utterances/s |")
    del g
