# Decode time against the batch size (T = 210): how far the team rounds are from "B = 128 at the latency of B = 32" (review item 7)
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp, decode_graph=0)
eng.set_decode_mode(3)
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
for B in (8, 16, 32, 64, 128):
    L = torch.from_numpy(synthetic_text(hp, B=B)).cuda()
    for _ in range(2): eng.text2mel(L)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3): eng.text_enc(L)
    t1.record(); torch.cuda.synchronize()
    te = t0.elapsed_time(t1) / 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): eng.text2mel(L)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"B {B:4d}: text2mel {ms:8.3f} ms, TextEnc {te:6.3f} ms, decode {(ms - te) * 1e3 / hp.max_T:7.2f} us per frame, {B * hp.max_T / ((ms - te) * 1e-3) / 1e3:8.1f} k mel frames/s (decode only)")
