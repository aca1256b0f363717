import os, sys, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_text
W = bench.cached_synthetic_weights(hp, 1234, 0, 1)
eng = Engine(W, hp, device=0, decode_graph=0)
L = torch.from_numpy(synthetic_text(hp, B=32, seed=1234)).cuda()
torch.cuda.synchronize()
print("before anything, default stream:", bench.timed(lambda: eng.text_enc(L)))
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
print("HP stream:", bench.timed(lambda: eng.text_enc(L)))
for _ in range(2): eng.synthesize(L); torch.cuda.synchronize(); eng.decode_status()
print("after warmup synth:", bench.timed(lambda: eng.text_enc(L)))
eng.prof_enable(bench.PROF_XCONE)
for _ in range(5): out = eng.synthesize(L)
torch.cuda.synchronize(); eng.prof_enable(-1); print(eng.prof_collect())
print("after timed region:", bench.timed(lambda: eng.text_enc(L)))
Z = out[1]
Zh = torch.empty(Z.shape, dtype=Z.dtype, pin_memory=True); Zh.copy_(Z); torch.cuda.synchronize()
print("after pinned gather:", bench.timed(lambda: eng.text_enc(L)))
print("text2mel:", bench.timed(lambda: eng.text2mel(L)), "ssrn:", bench.timed(lambda: eng.ssrn(out[0], want_logits=False)))
print("again:", bench.timed(lambda: eng.text_enc(L)), bench.timed(lambda: eng.text_enc(L), reps=20))
