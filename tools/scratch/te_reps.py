"""Why does bench.py's phases.textenc_ms read 2.2 ms when tools/ssrn_time.py reads 1.95?  TextEnc timed with 3 / 10 / 30 repetitions, from a default and from a high-priority stream."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_text, synthetic_weights
eng = Engine(synthetic_weights(hp, seed=1), hp)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
def t(fn, warm, n):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, st in (("default stream", torch.cuda.Stream()), ("high-priority stream", torch.cuda.Stream(priority=-1))):
    with torch.cuda.stream(st):
        for warm, n in ((1, 3), (3, 10), (3, 30)):
            eng.synthesize(L); torch.cuda.synchronize()
            print(f"{name}: warm {warm}, reps {n}: TextEnc {t(lambda: eng.text_enc(L), warm, n):.3f} ms")
