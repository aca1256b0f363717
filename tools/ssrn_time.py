"""SSRN phase time (B utterances, T = 210 -> (B, 840, 1025)) and TextEnc (B, 180), HIP-event timed on the caller's stream; env DCTTS_SSRN_SPLIT,
DCTTS_* knobs of tools/README.md apply."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import dc_tts_amd._lib as _L
if os.environ.get("DCTTS_AB_LIB"): _L.LIB_PATH = os.environ["DCTTS_AB_LIB"]      # A/B of two builds of the library (tools only; the product has no such switch)
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_text, synthetic_weights
eng = Engine(synthetic_weights(hp, seed=1), hp)
for B in [int(a) for a in sys.argv[1:]] or [32]:
    Y = torch.rand(B, 210, hp.n_mels, device="cuda")
    for _ in range(3): eng.ssrn(Y, want_logits=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    a.record()
    for _ in range(n): eng.ssrn(Y, want_logits=False)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"B={B}: SSRN {ms:.3f} ms, {B * 39.34e9 / ms / 1e9:.1f} TFLOP/s = {B * 39.34e9 / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
    L = torch.from_numpy(synthetic_text(hp, B=B)).cuda()
    for _ in range(3): eng.text_enc(L)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n): eng.text_enc(L)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"B={B}: TextEnc {ms:.3f} ms, {B * 2 * 3.0789e9 / ms / 1e9:.1f} TFLOP/s = {B * 2 * 3.0789e9 / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
