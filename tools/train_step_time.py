"""One training step of each network at the training batch (B = 32, T = 210, N = 180): wall time per step and the part of it the host
spends enqueueing (perf_counter around the calls, before the synchronize) -- says whether the step is launch/host-bound or GPU-bound."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.train import TrainGraph
from dc_tts_amd.weights import synthetic_weights
B, T, N = 32, 210, 180
Wn = synthetic_weights(hp, seed=1)
ids = torch.randint(1, len(hp.vocab), (B, N), dtype=torch.int32, device="cuda")
for num, name in ((1, "Text2Mel"), (2, "SSRN")):
    g = TrainGraph(num, Wn, hp)
    batch = (ids, torch.rand(B, T, hp.n_mels, device="cuda")) if num == 1 else \
        (torch.rand(B, T, hp.n_mels, device="cuda"), torch.rand(B, 4 * T, hp.n_linear, device="cuda"))
    for _ in range(2): g.train_op(*batch)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): g.train_op(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {1e3 * (t2 - t0) / n:.2f} ms/step, host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, {B * n / (t2 - t0):.0f} utterances/s")
    del g
