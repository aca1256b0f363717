"""Hyper-parameters of the DC-TTS synthesis path.

Mirrors the class attributes of the reference's ``hyperparams.py:7-47`` that the
synthesis path reads (signal constants, model widths, vocab, max_N / max_T, B).
The training-side fields (``prepro``, ``data``, ``test_data``, ``lr``, ``logdir``, ``sampledir``, ``num_iterations``:
hyperparams.py:10,35-47) are read by dc_tts_amd/data_load.py, prepo.py and train.py only.

``max_T`` is overridable (``replace(max_T=1000)``) because the long-form
configuration of BASELINE.json uses 1000 mel frames; the attention mask is
built from ``max_N`` / ``max_T`` exactly as ``networks.py:142-145`` does.
"""
from dataclasses import dataclass, replace as _replace


@dataclass(frozen=True)
class Hyperparams:
    # signal processing (hyperparams.py:13-24)
    sr: int = 22050
    n_fft: int = 2048
    frame_shift: float = 0.0125
    frame_length: float = 0.05
    n_mels: int = 80
    power: float = 1.5
    n_iter: int = 50
    preemphasis: float = 0.97
    max_db: int = 100
    ref_db: int = 20
    # model (hyperparams.py:27-32)
    r: int = 4
    dropout_rate: float = 0.05
    e: int = 128
    d: int = 256
    c: int = 512
    attention_win_size: int = 3
    # data (hyperparams.py:38-40)
    vocab: str = "PE abcdefghijklmnopqrstuvwxyz'.?"
    max_N: int = 180
    max_T: int = 210
    # pipeline / data (hyperparams.py:10,35-37)
    prepro: bool = True
    data: str = "/data/private/voice/LJSpeech-1.0"
    test_data: str = "harvard_sentences.txt"
    # training scheme (hyperparams.py:43-47)
    lr: float = 0.001
    logdir: str = "logdir/LJ01"
    sampledir: str = "samples"
    B: int = 32
    num_iterations: int = 2000000

    @property
    def hop_length(self) -> int:  # hyperparams.py:17 (int(22050*0.0125) = 275)
        return int(self.sr * self.frame_shift)

    @property
    def win_length(self) -> int:  # hyperparams.py:18
        return int(self.sr * self.frame_length)

    @property
    def n_linear(self) -> int:  # 1 + n_fft//2, networks.py:270
        return 1 + self.n_fft // 2

    @property
    def seconds_per_mel_frame(self) -> float:
        """Audio seconds represented by one reduced-rate mel frame (r linear frames)."""
        return self.r * self.hop_length / self.sr

    def replace(self, **kw) -> "Hyperparams":
        return _replace(self, **kw)


hp = Hyperparams()
