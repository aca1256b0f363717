"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/dctts_hip.h declares (no compute
calls without a GPU), weights container checks, layer tables, batch sharding (world_size-2 gloo), and the loud failure
of the product path when no GPU / library is present."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from dc_tts_amd import build
    return build.build(force=False, verbose=False)


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "dctts_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "dctts_hip_debug.h")).read()
    surface = set(re.findall(r"\b(dctts_[a-z0-9_]+)\s*\(", hdr))
    assert len(surface) >= 18 and not any("debug" in n or "prof" in n for n in surface)     # measurement hooks live in the debug header
    trn = open(os.path.join(ROOT, "include", "dctts_train.h")).read()
    train_syms = set(re.findall(r"\b(dctts_train_[a-z0-9_]+)\s*\(", trn))
    assert len(train_syms) == 20                                                             # the first training slice
    declared = surface | set(re.findall(r"\b(dctts_[a-z0-9_]+)\s*\(", dbg)) | train_syms
    lib = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported by libdctts_hip.so"
    from dc_tts_amd import _lib
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))      # ctypes table in sync with the header


def test_library_is_gfx950_code_object(built_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", built_lib], capture_output=True, text=True).stdout
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob and b"hconv_kernel" in blob and b"hsplit_kernel" in blob, out[:200]


def test_no_gpu_means_loud_failure(weights):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dc_tts_amd.engine import DcttsError, Engine
    with pytest.raises(DcttsError):
        Engine(weights)
    from dc_tts_amd import networks
    networks._engine = None
    with pytest.raises(RuntimeError):
        networks.TextEnc(None, training=False)


def test_vocoder_needs_gpu_and_validates_config():
    """No CPU fallback for the vocoder either; argument checks answer before any HIP call (status code + message over the ABI)."""
    import torch
    from dc_tts_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    bad = _lib.VocoderConfig(1024, 275, 1102, 50, 1.5, 0.97, 100.0, 20.0, 60.0, 2048, 512)        # n_fft the kernels are not built for
    assert lib.dctts_vocoder_create(ctypes.byref(h), 0, ctypes.byref(bad)) == -1 and "n_fft" in _lib.last_error()
    bad = _lib.VocoderConfig(2048, 2000, 1102, 50, 1.5, 0.97, 100.0, 20.0, 60.0, 2048, 512)       # hop > win
    assert lib.dctts_vocoder_create(ctypes.byref(h), 0, ctypes.byref(bad)) == -1 and "hop_length" in _lib.last_error()
    assert lib.dctts_vocoder_create(None, 0, None) == -1
    assert lib.dctts_spectrogram2wav(None, None, 1, 8, None, None, None) == -1
    assert lib.dctts_griffin_lim(None, None, 1, 8, 1, None, None, None) == -1
    if not torch.cuda.is_available():
        from dc_tts_amd.engine import DcttsError
        from dc_tts_amd.utils import Vocoder, spectrogram2wav
        with pytest.raises(DcttsError):
            Vocoder()
        with pytest.raises(DcttsError):
            spectrogram2wav(np.zeros((8, 1025), np.float32))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dc_tts_amd/ may import it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "dc_tts_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "dctts_ref" not in src and "vocoder_ref" not in src, os.path.join(dp, f)


def test_environment_knobs_are_the_documented_ones():
    """The library reads a handful of measurement / A-B knobs from the environment (once, when a context is created).  The set in the source and the
    table in tools/README.md must be the same, and small: knobs are how unmeasured variants pile up."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ""
    for name in ("dctts_api.hip", "decode_host.h"):
        src += open(os.path.join(root, "dc_tts_amd", "csrc", name)).read()
    in_source = set(re.findall(r'"(DCTTS_[A-Z_0-9]+)"', src))
    readme = open(os.path.join(root, "tools", "README.md")).read()
    table = readme.split("## Environment knobs")[1].split("Everything else")[0]
    documented = set(re.findall(r"`(DCTTS_[A-Z_0-9]+)`", table))
    assert in_source == documented, (sorted(in_source), sorted(documented))
    assert len(in_source) <= 8


def test_weights_container(weights):
    from dc_tts_amd.hyperparams import hp
    from dc_tts_amd.weights import check_weights, load_npz, save_npz, synthetic_text
    check_weights(weights, hp)
    bad = dict(weights); bad.pop("SSRN/C_16/conv1d/bias")
    with pytest.raises(ValueError):
        check_weights(bad, hp)
    bad = dict(weights); bad["SSRN/D_4/conv2d_transpose/kernel"] = np.zeros((1, 3, 512, 511), np.float32)
    with pytest.raises(ValueError):
        check_weights(bad, hp)
    assert weights["SSRN/D_4/conv2d_transpose/kernel"].shape == (1, 3, 512, 512)
    L = synthetic_text(hp, B=5, seed=1)
    assert L.shape == (5, hp.max_N) and L.dtype == np.int32
    for row in L:
        n = int((row != 0).sum())
        assert row[n - 1] == 1 and np.all(row[:n - 1] >= 2) and np.all(row[n:] == 0)       # ...chars, E, padding


def test_hyperparams_metric_constants():
    from dc_tts_amd.hyperparams import hp
    assert hp.hop_length == 275 and hp.n_linear == 1025 and len(hp.vocab) == 32
    assert abs(hp.seconds_per_mel_frame - 4 * 275 / 22050) < 1e-12
    assert hp.replace(max_T=1000).max_T == 1000 and hp.max_T == 210


def test_shard_bounds():
    from dc_tts_amd.sharding import shard_bounds
    for B in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(B, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(256, 8, 3) == (96, 128)                   # config 4: 8 x 32


_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dc_tts_amd.hyperparams import hp
from dc_tts_amd.sharding import synthesize_sharded
from dc_tts_amd.weights import synthetic_text, synthetic_weights
from oracle import dctts_ref as O
dist.init_process_group("gloo")
h = hp.replace(max_T=6)
W = synthetic_weights(h, seed=1234)
L = synthetic_text(h, B=3, seed=42)                         # ragged: ranks get 2 and 1 utterances
def synth(Ls):
    Y, Z, traj = O.synthesize(Ls, W, h, np.float32)
    return Y, Z, traj
out = synthesize_sharded(L, synth)
if dist.get_rank() == 0:
    Y, Z, traj = out
    Yr, Zr, tr = O.synthesize(L, W, h, np.float32)          # unsharded
    assert Y.shape == (3, 6, 80) and Z.shape == (3, 24, 1025)
    assert np.array_equal(traj, tr) and np.array_equal(Y, Yr) and np.array_equal(Z, Zr)
    print("SHARD_OK")
else:
    assert out is None
dist.destroy_process_group()
'''


def test_sharding_world2_gloo(tmp_path):
    """N > 1 path on CPU: two gloo ranks, sharded result == unsharded result, bitwise, in utterance order."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SHARD_OK" in r.stdout


def test_text_front_end_matches_reference_semantics(tmp_path):
    """dc_tts_amd.data_load (product) vs the oracle's restatement and hand-checked strings (data_load.py:19-31,79-86)."""
    from dc_tts_amd import data_load as D
    from dc_tts_amd.hyperparams import hp
    from oracle import dctts_ref as O
    c2i, i2c = D.load_vocab()
    assert c2i["P"] == 0 and c2i["E"] == 1 and c2i[" "] == 2 and i2c[31] == "?" and len(c2i) == 32
    assert D.text_normalize("Crème  BRÛLÉE, 42 times!") == "creme brulee times "
    lines = ["1. The birch canoe slid on the smooth planks.\n", "2. It's easy to tell the depth of a well.\n", "3. Déjà-vu?\n"]
    L = D.encode_lines(lines)
    np.testing.assert_array_equal(L, O.load_sentences(lines, hp))
    assert "".join(i2c[i] for i in L[1] if i) == "it's easy to tell the depth of a well.E"
    assert "".join(i2c[i] for i in L[2] if i) == "deja vu?E"
    f = tmp_path / "sent.txt"
    f.write_text("http://header.line\n" + "".join(lines), encoding="utf-8")
    np.testing.assert_array_equal(D.load_data("synthesize", str(f)), L)
    with pytest.raises(ValueError):
        D.load_data("validate")
    with pytest.raises(ValueError):
        D.encode_lines(["1. " + "a" * 200])


def test_wave_fft_lane_program_on_host(tmp_path):
    """dc_tts_amd/csrc/fft_wave.h is __host__ __device__: run the exact 64-lane program of the vocoder's FFT on the CPU
    (tests/fft_wave_test.cpp) against a double-precision DFT."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "fft_wave_test")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fft_wave_test.cpp")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", exe])
    fwd, inv = (float(x) for x in subprocess.check_output([exe]).decode().split())
    # inputs are uniform in [-1, 1): outputs have magnitude ~ sqrt(1024) * 0.8; 1e-4 absolute is ~4e-6 relative
    assert fwd < 1e-4 and inv < 1e-4, (fwd, inv)
