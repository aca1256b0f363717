"""Probe: can SSRN run on a CU partition while the (latency-bound) decode runs, and what does each pay?
SSRN alone (all CUs / masked), decode alone, both concurrently (SSRN on the masked stream)."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.engine import Engine, _ptr
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.weights import synthetic_weights, synthetic_text
eng = Engine(synthetic_weights(hp), hp)
lib = eng.lib
B, T = 32, 210
L = torch.from_numpy(synthetic_text(hp, B=B)).cuda()
Y, mx = eng.text2mel(L)
Z = torch.empty(B, 4 * T, hp.n_linear, device="cuda")
Y2 = torch.empty_like(Y); mx2 = torch.empty_like(mx)
def ssrn_on(stream_ptr):
    rc = lib.dctts_ssrn_fwd(eng._h, _ptr(Y), B, T, None, _ptr(Z), stream_ptr)
    assert rc == 0, rc
def decode():
    rc = lib.dctts_text2mel_decode(eng._h, _ptr(L), B, hp.max_N, T, _ptr(Y2), _ptr(mx2), eng._stream())
    assert rc == 0, rc
def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("decode alone ms", round(wall(decode), 2))
print("ssrn alone (all CUs) ms", round(wall(lambda: ssrn_on(eng._stream())), 2))
for first, count in ((192, 64), (128, 128), (224, 32)):
    s = ctypes.c_void_p()
    assert lib.dctts_debug_stream_create(first, count, ctypes.byref(s)) == 0
    t_s = wall(lambda: ssrn_on(s))
    def both():
        ssrn_on(s); decode()
    t_b = wall(both)
    print(f"mask CUs [{first},{first+count}): ssrn alone {t_s:.2f} ms; ssrn || decode wall {t_b:.2f} ms")
    torch.cuda.synchronize()
    lib.dctts_debug_stream_destroy(s)
