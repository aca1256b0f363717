"""Host driver shaped like the reference's ``synthesize.py:21-57`` (without Griffin-Lim / wav I/O,
which are outside the metric).

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
    """L (B, max_N) int32 on the engine's GPU -> (Y (B,T,80), Z (B,4T,1025), max_attentions (B,T))."""
    eng = engine or networks.bound_engine()
    return eng.synthesize(L, max_T)


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
    = the reference's `python synthesize.py` up to the magnitude spectrograms (synthesize.py:21-57): the mel (Y) and linear
    (Z) spectrograms are written as .npy per sentence; Griffin-Lim / wav writing (synthesize.py:59-64) is out of scope."""
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
    args = ap.parse_args(argv)
    if args.logdir:
        from .tf_checkpoint import load_reference_weights
        W = load_reference_weights(args.logdir, hp)
    else:
        W = synthetic_weights(hp, seed=1234, perturb=True)
    eng = Engine(W, hp)
    L = load_data("synthesize", args.text, hp)
    Y, Z, _ = eng.synthesize(torch.from_numpy(L).to(eng.device))
    os.makedirs(args.out, exist_ok=True)
    for i in range(L.shape[0]):
        np.save(os.path.join(args.out, f"{i + 1}.mel.npy"), Y[i].cpu().numpy())
        np.save(os.path.join(args.out, f"{i + 1}.mag.npy"), Z[i].cpu().numpy())
    print(f"{L.shape[0]} sentences -> {args.out}/<n>.mel.npy ({tuple(Y.shape[1:])}), <n>.mag.npy ({tuple(Z.shape[1:])})")


if __name__ == "__main__":
    main()
