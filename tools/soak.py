"""Soak run of the decode's in-launch hand-offs (xgroup_kernel / xcone_kernel meet through bounded waits): how often does a decode fail?

  phase A   N1 decodes at B = 32, T = 210 back to back on one stream
  phase B   N2 decodes at B = 134 (team rounds: more utterance groups than teams)
  phase C   N3 decodes at B = 32 while ANOTHER stream of the same process runs SSRN + the Griffin-Lim vocoder of the "previous batch"
            (unrelated kernels competing for CUs, L2 and the fabric)
  phase D   N4 rounds of 2 x SSRN, 2 x TextEnc and 2 x synthesize of DIFFERENT inputs enqueued on two streams of the one engine (round 5: same-kind calls are
            ordered by the context's use groups): every output compared bitwise with the solo run of its input
  phase E   (round 6) N5 decodes each at B = 8, 16, 5: teams of one / two utterances per round

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


def phase_d(eng, n, every, T=60):
    h = hp.replace(max_T=T)
    La, Lb = (torch.from_numpy(synthetic_text(h, B=32, seed=s)).cuda() for s in (61, 62))
    Ma, Mb = torch.rand(32, T, hp.n_mels, device="cuda"), torch.rand(32, T, hp.n_mels, device="cuda")
    solo = {}
    for tag, L, M in (("a", La, Ma), ("b", Lb, Mb)):
        solo[tag] = (eng.text_enc(L), eng.ssrn(M, want_logits=False)[1], eng.synthesize(L, max_T=T))
    eng.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    mism = torch.zeros((), dtype=torch.int64, device="cuda")
    reports = []
    t0 = time.time()
    for i in range(n):
        order = ((s1, "a", La, Ma), (s2, "b", Lb, Mb)) if i % 2 == 0 else ((s2, "b", Lb, Mb), (s1, "a", La, Ma))
        for what in (0, 1, 2):
            for st, tag, L, M in order:
                with torch.cuda.stream(st):
                    try:
                        if what == 0: got = eng.text_enc(L)
                        elif what == 1: got = (eng.ssrn(M, want_logits=False)[1],)
                        else: got = eng.synthesize(L, max_T=T)
                    except DcttsError as e:
                        reports.append("refused: " + str(e)[:160]); continue
                    ref = solo[tag][what] if what != 1 else (solo[tag][1],)
                    bad = torch.zeros((), dtype=torch.bool, device="cuda")
                    for x, y in zip(got, ref): bad = bad | (x != y).any()
                    mism_local = bad.to(torch.int64)
                    # the counter lives on the default stream: add the flag there once this stream's work is done
                    ev = torch.cuda.Event(); ev.record(st)
                torch.cuda.current_stream().wait_event(ev)
                mism += mism_local
        if (i + 1) % every == 0 or i + 1 == n:
            torch.cuda.synchronize()
            try:
                eng.decode_status()
            except DcttsError as e:
                reports.append(f"after {i + 1} rounds: {str(e)[:200]}")
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"D (2 x SSRN + 2 x TextEnc + 2 x synthesize of different inputs on two streams): {n} rounds, B = 32, T = {T}: {dt:.1f} s wall ({1e3 * dt / n:.2f} ms per round); "
          f"status reports {len(reports)}, outputs that differ from their solo run {int(mism)}")
    for r in reports[:10]:
        print("   ", r)
    return len(reports) + int(mism)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=5000)
    ap.add_argument("--n2", type=int, default=500)
    ap.add_argument("--n3", type=int, default=1500)
    ap.add_argument("--n4", type=int, default=1000)
    ap.add_argument("--n5", type=int, default=1000)
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
    bad += phase_d(eng, a.n4, a.every)
    # phase E (round 6): the batch-sized teams -- one / two utterances per team and round (B <= 8 / 16)
    if a.n5 > 0:
        for B, T in ((8, hp.max_T), (16, 120), (5, 90)):
            bad += phase(eng, f"E (teams sized by the batch, B = {B})", B, a.n5, a.every, T)
    print("TOTAL failures:", bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
