"""The reference's data_load.py: the text front-end of the synthesis path (`load_vocab` data_load.py:19-22, `text_normalize`
:24-31, `load_data("synthesize")` :79-86) and the training input side (`load_data("train")` :33-77, `get_batch` :88-140).

Host-side, tiny; needed so a literal synthesize.py-style driver drops in.  Same names and behaviour:
accents stripped (NFD, category Mn removed), lower-cased, every character outside hp.vocab -> space, runs of spaces
squeezed; each line's leading "<n>. " numbering removed with split(" ", 1)[-1]; "E" (EOS) appended; zero ("P") padded to
hp.max_N; the first line of the file is a header and is skipped.
"""
import codecs
import os
import re
import unicodedata
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from .hyperparams import Hyperparams, hp as _hp


def load_vocab(hp: Hyperparams = _hp) -> Tuple[Dict[str, int], Dict[int, str]]:
    char2idx = {char: idx for idx, char in enumerate(hp.vocab)}
    idx2char = {idx: char for idx, char in enumerate(hp.vocab)}
    return char2idx, idx2char


def text_normalize(text: str, hp: Hyperparams = _hp) -> str:
    text = "".join(char for char in unicodedata.normalize("NFD", text) if unicodedata.category(char) != "Mn")
    text = text.lower()
    text = re.sub("[^{}]".format(hp.vocab), " ", text)
    text = re.sub("[ ]+", " ", text)
    return text


def encode_lines(lines: Iterable[str], hp: Hyperparams = _hp) -> np.ndarray:
    """Lines of a test file (header already removed) -> (len(lines), max_N) int32, data_load.py:81-86."""
    char2idx, _ = load_vocab(hp)
    sents = [text_normalize(line.split(" ", 1)[-1], hp).strip() + "E" for line in lines]
    texts = np.zeros((len(sents), hp.max_N), np.int32)
    for i, sent in enumerate(sents):
        if len(sent) > hp.max_N:
            raise ValueError(f"sentence {i} has {len(sent)} symbols > max_N={hp.max_N} "
                             "(the reference would raise a numpy broadcast error here)")
        texts[i, :len(sent)] = [char2idx[char] for char in sent]
    return texts


def load_data(mode: str = "train", path: Optional[str] = None, hp: Hyperparams = _hp):
    """data_load.py:33-86.  mode "train": (fpaths, text_lengths, texts) parsed from `<hp.data>/transcript.csv` (`path` overrides the
    file) -- LJ Speech rows `fname|raw|normalised`, any other corpus (the reference's "nick or kate") `fname|_|text|_|duration` with
    utterances over 10 s dropped; texts are int32 id arrays ending in E (the reference carries them as raw bytes for its TF queue).
    mode "synthesize": parse `path` (default hp.test_data), skipping its header line -> (n, max_N) int32."""
    if mode == "synthesize":
        lines = codecs.open(path or hp.test_data, "r", "utf-8").readlines()[1:]
        return encode_lines(lines, hp)
    if mode != "train":
        raise ValueError("load_data: mode is 'train' or 'synthesize'")
    char2idx, _ = load_vocab(hp)
    fpaths, text_lengths, texts = [], [], []
    lines = codecs.open(path or os.path.join(hp.data, "transcript.csv"), "r", "utf-8").readlines()
    for line in lines:
        if "LJ" in hp.data:
            fname, _, text = line.strip().split("|")
            fpath = os.path.join(hp.data, "wavs", fname + ".wav")
            text = text_normalize(text, hp) + "E"
        else:
            fname, _, text, _, duration = line.strip().split("|")
            if float(duration) > 10.0:
                continue
            fpath = os.path.join(hp.data, fname)
            text += "E"
        fpaths.append(fpath)
        ids = np.array([char2idx[char] for char in text], np.int32)
        text_lengths.append(len(ids))
        texts.append(ids)
    return fpaths, text_lengths, texts


def _read_spectrograms(fpath: str, hp: Hyperparams, prepro_dir: str):
    """data_load.py:107-116: the arrays `python -m dc_tts_amd.prepo` wrote (hp.prepro) or straight from the wave file."""
    fname = os.path.basename(fpath)
    if hp.prepro:
        return fname, np.load(os.path.join(prepro_dir, "mels", fname.replace("wav", "npy"))), \
            np.load(os.path.join(prepro_dir, "mags", fname.replace("wav", "npy")))
    from .audio import load_spectrograms
    return load_spectrograms(fpath, hp)


def _pad_stack(arrays: List[np.ndarray], multiple: int = 1) -> np.ndarray:
    """dynamic_pad=True: zero-pad axis 0 of every array to the longest in the batch (rounded up to `multiple`), stack."""
    n = max(a.shape[0] for a in arrays)
    n = (n + multiple - 1) // multiple * multiple
    out = np.zeros((len(arrays), n) + arrays[0].shape[1:], arrays[0].dtype)
    for i, a in enumerate(arrays):
        out[i, :a.shape[0]] = a
    return out


class BatchQueue:
    """`get_batch()` (data_load.py:88-140) as a Python iterator: an endless, per-epoch reshuffled stream of utterances
    (tf.train.slice_input_producer(shuffle=True)) is routed into buckets by text length with boundaries
    range(minlen + 1, maxlen - 1, 20) (bucket i takes boundaries[i-1] <= len < boundaries[i]); a bucket that holds hp.B utterances is
    emitted as one batch, every tensor zero-padded to the longest member (tf.contrib.training.bucket_by_sequence_length(dynamic_pad=True)).
    Yields (texts (B, N) int32, mels (B, T / r, n_mels), mags (B, T, 1 + n_fft / 2), fnames).  Leftovers stay in their bucket for the
    next epoch, as in a queue.  The reference's 8 reader threads make its batch order nondeterministic; here it is a function of `seed`.
    `pad_text_to`: N is rounded up to this multiple with more P (id 0) columns (default 1 = the reference's padding to the longest text of the batch;
    the training attention takes any N)."""

    def __init__(self, hp: Hyperparams = _hp, seed: int = 0, prepro_dir: str = ".", transcript: Optional[str] = None, pad_text_to: int = 1):
        self.hp, self.prepro_dir, self.pad_text_to = hp, prepro_dir, pad_text_to
        self.fpaths, self.text_lengths, self.texts = load_data("train", transcript, hp)
        if not self.fpaths:
            raise ValueError("get_batch: the transcript lists no utterance")
        maxlen, minlen = max(self.text_lengths), min(self.text_lengths)
        self.num_batch = len(self.fpaths) // hp.B                                                        # data_load.py:97
        self.boundaries = list(range(minlen + 1, maxlen - 1, 20))
        self.buckets: List[List[int]] = [[] for _ in range(len(self.boundaries) + 1)]
        self.rng = np.random.default_rng(seed)

    def which_bucket(self, length: int) -> int:
        return int(np.searchsorted(self.boundaries, length, side="right"))

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, List[str]]]:
        while True:
            for idx in self.rng.permutation(len(self.fpaths)):
                b = self.buckets[self.which_bucket(self.text_lengths[idx])]
                b.append(int(idx))
                if len(b) == self.hp.B:
                    items = [_read_spectrograms(self.fpaths[i], self.hp, self.prepro_dir) for i in b]
                    texts = _pad_stack([self.texts[i] for i in b], self.pad_text_to)
                    yield texts, _pad_stack([m for _, m, _ in items]), _pad_stack([g for _, _, g in items]), [f for f, _, _ in items]
                    b.clear()


def get_batch(hp: Hyperparams = _hp, **kw):
    """texts, mels, mags, fnames come from iterating the returned queue; `.num_batch` is data_load.py:97's count."""
    return BatchQueue(hp, **kw)
