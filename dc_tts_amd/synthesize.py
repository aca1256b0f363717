"""Host driver shaped like the reference's ``synthesize.py:21-64``.  The metric covers :45-57 (decode + SSRN);
the Griffin-Lim tail (:59-64) runs on the GPU too (``dc_tts_amd/utils.py``) and is used by the CLI below.

``synthesize_reference_loop`` is the literal loop of synthesize.py:45-57 written against the drop-in
``networks`` surface -- 210 full-graph evaluations, exactly what a maintainer gets by swapping the
import; ``synthesize`` is the fast path (one C-ABI call: incremental exact decode + SSRN).
Both produce the same (Y, Z); tests/test_gpu_parity.py checks them against each other and the oracle.
"""
from typing import Optional

import torch

from . import networks
from .engine import Engine


def synthesize(L: torch.Tensor, engine: Optional[Engine] = None, max_T: Optional[int] = None):
    """L (B, max_N) int32 on the engine's GPU -> (Y (B,T,80), Z (B,4T,1025), max_attentions (B,T)).
    Like the `sess.run` calls it replaces this returns checked results: it waits for the decode and, should it have failed on the
    device (the GPU shared with another process's kernels), repeats it once with one launch per layer (`Engine.synthesize(check=True)`)."""
    eng = engine or networks.bound_engine()
    return eng.synthesize(L, max_T, check=True)


def synthesize_reference_loop(L: torch.Tensor, engine: Optional[Engine] = None):
    """synthesize.py:45-57 verbatim on the GPU network functions (O(T^2) work; for parity checks)."""
    eng = engine or networks.bound_engine()
    hp = eng.hp
    B = L.shape[0]
    Y = torch.zeros(B, hp.max_T, hp.n_mels, device=eng.device)                       # :45
    prev_max_attentions = torch.zeros(B, dtype=torch.int32, device=eng.device)       # :46
    K, V = eng.text_enc(L)            # pure function of L: the reference recomputes it each step
    traj = torch.zeros(B, hp.max_T, dtype=torch.int64, device=eng.device)
    for j in range(hp.max_T):                                                        # :47
        S = torch.cat((torch.zeros_like(Y[:, :1, :]), Y[:, :-1, :]), 1).contiguous()  # train.py:51
        Q = eng.audio_enc(S)
        R, _al, max_att = eng.attention(Q, K, V, True, prev_max_attentions)
        _logits, _Y = eng.audio_dec(R)
        Y[:, j, :] = _Y[:, j, :]                                                     # :53
        prev_max_attentions = max_att[:, j].to(torch.int32).contiguous()             # :54
        traj[:, j] = max_att[:, j]
    _zl, Z = eng.ssrn(Y, want_logits=False)                                          # :57
    return Y, Z, traj


def main(argv=None):
    """`python -m dc_tts_amd.synthesize --text harvard_sentences.txt [--logdir logdir/LJ01] --out samples`
    = the reference's `python synthesize.py` (synthesize.py:21-64): Text2Mel decode, SSRN, then `spectrogram2wav` per sentence
    (on the GPU, dc_tts_amd/utils.py) written as `<out>/<n>.wav` with scipy.io.wavfile.write like synthesize.py:64.
    `--save-spectrograms` also writes the mel (Y) and linear (Z) spectrograms as .npy."""
    import argparse
    import os

    import numpy as np

    from .data_load import load_data
    from .hyperparams import hp
    from .weights import synthetic_weights
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--text", required=True, help="test file in the reference's format (header line, then '<n>. sentence')")
    ap.add_argument("--logdir", default=None, help="prefix of the trained checkpoints (<logdir>-1 Text2Mel, <logdir>-2 SSRN); "
                                                   "omitted: seeded synthetic weights")
    ap.add_argument("--out", default="samples")
    ap.add_argument("--save-spectrograms", action="store_true")
    ap.add_argument("--no-wav", action="store_true", help="stop at the magnitude spectrograms (synthesize.py:57)")
    args = ap.parse_args(argv)
    if args.logdir:
        from .tf_checkpoint import load_reference_weights
        W = load_reference_weights(args.logdir, hp)
    else:
        W = synthetic_weights(hp, seed=1234, perturb=True)
    eng = Engine(W, hp)
    L = load_data("synthesize", args.text, hp)
    Y, Z, _ = eng.synthesize(torch.from_numpy(L).to(eng.device), check=True)     # waits, checks the decode's status, repeats it once if it failed
    os.makedirs(args.out, exist_ok=True)
    if args.save_spectrograms or args.no_wav:
        for i in range(L.shape[0]):
            np.save(os.path.join(args.out, f"{i + 1}.mel.npy"), Y[i].cpu().numpy())
            np.save(os.path.join(args.out, f"{i + 1}.mag.npy"), Z[i].cpu().numpy())
    if not args.no_wav:
        from scipy.io.wavfile import write
        from .utils import spectrogram2wav
        for i, wav in enumerate(spectrogram2wav(Z, hp)):                # synthesize.py:61-64
            write(os.path.join(args.out, f"{i + 1}.wav"), hp.sr, wav)
    print(f"{L.shape[0]} sentences -> {args.out}/ (mel {tuple(Y.shape[1:])}, mag {tuple(Z.shape[1:])}"
          f"{'' if args.no_wav else ', <n>.wav at %d Hz' % hp.sr})")


if __name__ == "__main__":
    main()
