"""In-tree build of libdctts_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", "dctts_api.hip"), os.path.join(HERE, "csrc", "vocoder_api.hip"), os.path.join(HERE, "csrc", "train_api.hip")]
OUT = os.path.join(HERE, "lib", "libdctts_hip.so")


def _deps():
    return SRCS + glob.glob(os.path.join(HERE, "csrc", "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in _deps()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + SRCS + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
