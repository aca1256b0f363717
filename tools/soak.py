"""Soak run of the decode's in-launch hand-offs (xgroup_kernel / xcone_kernel meet through bounded waits): how often does a decode fail?

  phase A   N1 decodes at B = 32, T = 210 back to back on one stream
  phase B   N2 decodes at B = 134 (team rounds: more utterance groups than teams)
  phase C   N3 decodes at B = 32 while ANOTHER stream of the same process runs SSRN + the Griffin-Lim vocoder of the "previous batch"
            (unrelated kernels competing for CUs, L2 and the fabric)

Every decode's outputs are compared ON THE DEVICE with the first decode of its phase (bitwise: the decode is deterministic); the status word is
read every `--every` decodes.  A failed decode shows up three ways -- the status report, NaN outputs, a mismatch count -- and all three are printed.

    python tools/soak.py [--n1 5000 --n2 500 --n3 1500] > profiles/r04_soak.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dc_tts_amd.engine import DcttsError, Engine            # noqa: E402
from dc_tts_amd.hyperparams import hp                       # noqa: E402
from dc_tts_amd.weights import synthetic_text, synthetic_weights  # noqa: E402


def phase(eng, name, B, n, every, T, side=None):
    h = hp.replace(max_T=T)
    L = torch.from_numpy(synthetic_text(h, B=B, seed=1234)).cuda()
    Y0, m0 = eng.text2mel(L, max_T=T)
    eng.synchronize()
    assert bool(torch.isfinite(Y0).all())
    mism = torch.zeros((), dtype=torch.int64, device="cuda")
    nans = torch.zeros((), dtype=torch.int64, device="cuda")
    reports, refused = [], 0
    t0 = time.time()
    done = 0
    while done < n:
        k = min(every, n - done)
        for _ in range(k):
            if side is not None:
                side()
            try:
                Y, m = eng.text2mel(L, max_T=T)
            except DcttsError as e:                            # the one refusal after an unreported failure
                refused += 1
                reports.append("refused: " + str(e)[:160])
                continue
            bad = (Y != Y0).any() | (m != m0).any()
            mism += bad.to(torch.int64)
            nans += torch.isnan(Y).any().to(torch.int64)
        done += k
        try:
            eng.synchronize()
        except DcttsError as e:
            reports.append(f"after {done} decodes: {str(e)[:200]}")
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{name}: {n} decodes, B = {B}, T = {T}: {dt:.1f} s wall ({1e3 * dt / n:.2f} ms per decode, {1e6 * dt / n / T:.1f} us per frame incl. the checks); "
          f"status reports {len(reports)}, refused calls {refused}, decodes with NaN outputs {int(nans)}, decodes that differ from the first {int(mism)}")
    for r in reports[:10]:
        print("   ", r)
    return len(reports) + int(nans) + int(mism)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=5000)
    ap.add_argument("--n2", type=int, default=500)
    ap.add_argument("--n3", type=int, default=1500)
    ap.add_argument("--every", type=int, default=250)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    W = synthetic_weights(hp, seed=1234, perturb=True)
    eng = Engine(W, hp)
    print(f"# soak of the decode hand-offs on {torch.cuda.get_device_name(0)}; team kernels on, eager launches; every decode compared bitwise with its phase's first")
    bad = 0
    bad += phase(eng, "A (back to back)", 32, a.n1, a.every, hp.max_T)
    bad += phase(eng, "B (team rounds)", 134, a.n2, a.every, 100)
    # phase C: SSRN + vocoder of a previous batch on another stream of this process
    from dc_tts_amd.utils import Vocoder
    voc = Vocoder(hp.replace(n_iter=4), device=0)
    s2 = torch.cuda.Stream()
    Yprev = torch.rand(32, hp.max_T, hp.n_mels, device="cuda")
    state = {"i": 0}

    def side():
        if state["i"] % 2 == 0:                                # one SSRN pass (10 ms) + 4 Griffin-Lim iterations per two decodes (2 x 21 ms): the other stream is busy about a third of the time
            with torch.cuda.stream(s2):
                Z = eng.ssrn(Yprev, want_logits=False)[1]
                voc.spectrogram2wav_device(Z)
        state["i"] += 1
    bad += phase(eng, "C (SSRN + vocoder on a second stream)", 32, a.n3, a.every, hp.max_T, side)
    print("TOTAL failures:", bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
