# Decode time against the number of frames: slope = steady-state cost of a frame, intercept = what a decode costs besides its frames
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp, decode_graph=int(os.environ.get("GM", "0")))
eng.set_decode_mode(3)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
res = {}
for T in (420, 210, 120):
    for _ in range(2): eng.text2mel(L, max_T=T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.text2mel(L, max_T=T)
    e1.record(); torch.cuda.synchronize()
    res[T] = e0.elapsed_time(e1) / 5
    print("T", T, "text2mel ms", res[T])
print("slope 210..420: %.2f us/frame; 120..210: %.2f us/frame; intercept (210 line): %.3f ms" % ((res[420] - res[210]) / 210 * 1e3, (res[210] - res[120]) / 90 * 1e3, res[210] - (res[420] - res[210])))
