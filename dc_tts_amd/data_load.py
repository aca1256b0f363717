"""Text front-end of the synthesis path: the `synthesize` branch of the reference's data_load.py
(`load_vocab` data_load.py:19-22, `text_normalize` :24-31, `load_data("synthesize")` :79-86).

Host-side, tiny; needed so a literal synthesize.py-style driver drops in.  Same names and behaviour:
accents stripped (NFD, category Mn removed), lower-cased, every character outside hp.vocab -> space, runs of spaces
squeezed; each line's leading "<n>. " numbering removed with split(" ", 1)[-1]; "E" (EOS) appended; zero ("P") padded to
hp.max_N; the first line of the file is a header and is skipped.
"""
import codecs
import re
import unicodedata
from typing import Dict, Iterable, Tuple

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


def load_data(mode: str = "synthesize", path: str = "harvard_sentences.txt", hp: Hyperparams = _hp) -> np.ndarray:
    """`load_data("synthesize")`: parse `path` (hp.test_data in the reference), skipping its header line."""
    if mode != "synthesize":
        raise NotImplementedError("only the synthesize branch of load_data is on the synthesis path (training data loading "
                                  "is out of scope)")
    lines = codecs.open(path, "r", "utf-8").readlines()[1:]
    return encode_lines(lines, hp)
