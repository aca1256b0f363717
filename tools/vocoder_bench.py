"""Vocoder timing workload: spectrogram2wav for one bench batch (B=32, F=840, n_iter=50), HIP-event timed.
Also the workload for `rocprofv3 --kernel-trace --stats` / `--pmc` of the Griffin-Lim kernels."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.utils import Vocoder
B = int(os.environ.get("VB", 32)); F = int(os.environ.get("VF", 840)); reps = int(os.environ.get("VREPS", 5))
v = Vocoder(hp)
mag = torch.rand(B, F, hp.n_linear, device="cuda")
for _ in range(2):
    v.spectrogram2wav_device(mag)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    wav, b = v.spectrogram2wav_device(mag)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
audio_s = B * hp.hop_length * (F - 1) / hp.sr
alg = B * F * (2 * hp.win_length + hp.n_linear) * 4
print(json.dumps({"B": B, "F": F, "n_iter": hp.n_iter, "ms_per_batch": round(ms, 3), "audio_seconds": round(audio_s, 2),
                  "rtf": ms * 1e-3 / audio_s, "us_per_gl_iteration": round(ms * 1e3 / hp.n_iter, 2),
                  "gl_iter_algorithmic_bytes": alg, "gl_iter_GBps_if_all_time": round(alg / (ms * 1e-3 / hp.n_iter) / 1e9, 1),
                  "device_bytes": v.device_bytes()}))
