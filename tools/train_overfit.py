"""Sanity run of the training step: TrainGraph.train_op repeated on ONE fixed synthetic batch (no data pipeline exists), starting at the
peak of the Noam schedule.  The losses must fall: L1 / binary divergence towards the batch's targets, guided attention towards the
diagonal.  Prints the trajectory (profiles/r02_train_overfit.txt)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.train import TrainGraph
from dc_tts_amd.weights import synthetic_weights
rng = np.random.default_rng(0)
W = synthetic_weights(hp, seed=3)
B, N, T = 8, 40, 60
ids = torch.from_numpy(rng.integers(2, len(hp.vocab), (B, N)).astype(np.int32)).cuda()
# a smooth, learnable target: a few drifting spectral bumps per utterance
t = np.arange(T)[None, :, None]; f = np.arange(hp.n_mels)[None, None, :]
c = rng.uniform(10, 70, (B, 1, 1)) + rng.uniform(-0.3, 0.3, (B, 1, 1)) * t
mels = torch.from_numpy((0.1 + 0.8 * np.exp(-((f - c) / 6.0) ** 2)).astype(np.float32)).cuda()
fl = np.arange(hp.n_linear)[None, None, :]; tl = np.arange(4 * T)[None, :, None]
cl = (rng.uniform(100, 900, (B, 1, 1)) + rng.uniform(-1, 1, (B, 1, 1)) * tl)
mags = torch.from_numpy((0.1 + 0.8 * np.exp(-((fl - cl) / 60.0) ** 2)).astype(np.float32)).cuda()
for num, batch, steps, names in ((1, (ids, mels), 300, "loss_mels loss_bd1 loss_att"), (2, (mels, mags), 120, "loss_mags loss_bd2")):
    g = TrainGraph(num, W, hp, training=True, seed=1)
    g.global_step = 3999
    print(f"# Graph(num={num}): {steps} steps on one fixed batch (B={B}, T={T}{', N=' + str(N) if num == 1 else ''}), dropout {hp.dropout_rate}, lr from utils.py:142-145 at global_step 3999+")
    print("# step  " + names + "   ms/step")
    t0 = time.perf_counter()
    for s in range(steps + 1):
        losses = g.train_op(*batch)
        if s % (steps // 10) == 0:
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (s + 1) * 1e3
            print(f"{s:5d}  " + "  ".join(f"{v:.5f}" for v in losses.cpu().tolist()) + f"   {dt:.1f}")
    del g
