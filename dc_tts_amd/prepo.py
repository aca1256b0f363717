"""prepo.py of the reference: every utterance of the training transcript -> `mels/<name>.npy` (T / r, n_mels) and `mags/<name>.npy`
(T, 1 + n_fft / 2), once, so that the training loop reads arrays instead of decoding audio (hp.prepro, data_load.py:107-114).

    python -m dc_tts_amd.prepo [--data <corpus dir>] [--out <dir holding mels/ and mags/>, default .] [--workers N]

The reference runs the utterances one after the other; they are independent, so `--workers` spreads them over processes."""
import argparse
import os
import sys
from typing import List, Optional

import numpy as np

from .audio import load_spectrograms
from .data_load import load_data
from .hyperparams import Hyperparams, hp as _hp


def _one(job):
    fpath, out, hp = job
    fname, mel, mag = load_spectrograms(fpath, hp)
    np.save(os.path.join(out, "mels", fname.replace("wav", "npy")), mel)
    np.save(os.path.join(out, "mags", fname.replace("wav", "npy")), mag)
    return fname


def main(argv: Optional[List[str]] = None, hp: Hyperparams = _hp) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data", default=None)
    ap.add_argument("--out", default=".")
    ap.add_argument("--workers", type=int, default=1)
    args = ap.parse_args(argv)
    if args.data:
        hp = hp.replace(data=args.data)
    fpaths, _, _ = load_data("train", hp=hp)
    for sub in ("mels", "mags"):
        os.makedirs(os.path.join(args.out, sub), exist_ok=True)
    jobs = [(f, args.out, hp) for f in fpaths]
    if args.workers > 1:
        from multiprocessing import Pool
        with Pool(args.workers) as pool:
            for i, _ in enumerate(pool.imap_unordered(_one, jobs, chunksize=8)):
                if (i + 1) % 500 == 0: print(f"{i + 1}/{len(jobs)}", file=sys.stderr)
    else:
        for job in jobs:
            _one(job)
    return 0


if __name__ == "__main__":
    sys.exit(main())
