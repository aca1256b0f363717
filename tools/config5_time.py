"""BASELINE configs[4]'s share of one GPU: max_T = 1000, B = 8 -- decode only and the full synthesis, event-timed.  XG=2 forces four utterances per team (round 5's dealing)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
if os.environ.get("XG"): os.environ["DCTTS_XGROUP"] = os.environ["XG"]
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
T, B = int(os.environ.get("T5", "1000")), int(os.environ.get("B5", "8"))
h = hp.replace(max_T=T)
eng = Engine(synthetic_weights(h, seed=1234, perturb=True), h)
L = torch.from_numpy(synthetic_text(h, B=B, seed=77)).cuda()
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
te = timed(lambda: eng.text_enc(L), 5)
t2m = timed(lambda: eng.text2mel(L))
full = timed(lambda: eng.synthesize(L))
eng.decode_status()
print(f"B={B} T={T} XGROUP={os.environ.get('DCTTS_XGROUP', '1')}: TextEnc {te:.3f} ms, decode {(t2m - te) * 1e3 / T:.2f} us/frame, full {full:.2f} ms = {B * T / full:.1f} k mel frames/s")
