"""Per-kernel register / LDS / scratch table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (CPU only: no GPU needed).
usage: python tools/kernel_resources.py [source.hip ...]   (default: the three translation units of libdctts_hip.so)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = sys.argv[1:] or [os.path.join(ROOT, "dc_tts_amd", "csrc", n) for n in ("dctts_api.hip", "vocoder_api.hip", "train_api.hip")]
KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]")]
for src in SRCS:
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.path.join(d, "o.o"),
                            "-Rpass-analysis=kernel-resource-usage", "--offload-device-only"] + os.environ.get("KR_FLAGS", "").split(),
                           capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(1)
    print("#", os.path.relpath(src, ROOT))
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split("\n")[0].strip().split(" ")[0]
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(dctts::\w+Params.*", "", dn).replace("void dctts::", "")
        vals = []
        for k, pat in KEYS:
            m = re.search(pat + r": (\S+)", b)
            vals.append(f"{k}={m.group(1) if m else '?'}")
        print(f"{dn[:70]:70s} " + " ".join(vals))
