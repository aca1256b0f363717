"""text2mel -> ssrn back to back as synthesize() runs them, HIP events between the phases (no host synchronisation inside a step)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp, seed=1), hp)
L = torch.from_numpy(synthetic_text(hp, B=32)).cuda()
ev = lambda: torch.cuda.Event(enable_timing=True)
for _ in range(2):
    Y, _ = eng.text2mel(L); eng.ssrn(Y, want_logits=False)
torch.cuda.synchronize()
n = 5
E = [[ev(), ev(), ev()] for _ in range(n)]
for i in range(n):
    E[i][0].record(); Y, _ = eng.text2mel(L); E[i][1].record(); eng.ssrn(Y, want_logits=False); E[i][2].record()
torch.cuda.synchronize()
t2m = sum(e[0].elapsed_time(e[1]) for e in E) / n; ss = sum(e[1].elapsed_time(e[2]) for e in E) / n
tot = E[0][0].elapsed_time(E[-1][2]) / n
print(f"in situ: text2mel {t2m:.3f} ms, ssrn {ss:.3f} ms, step {tot:.3f} ms")
