"""Build-time guard for the decode's team kernels (CPU only: hipcc cross-compiles for gfx950 and reports every kernel's resources).

What the round-5 work on these kernels kept running into: a kernel that lives near the 256-register limit starts to SPILL after an innocent edit
(xcone_kernel at 249 registers, the fold's 600 bytes of scratch), and any scratch use costs ~5 us per launch on a path where a frame is ~76 us.
So: no scratch, and at most one workgroup's worth of registers / LDS per compute unit for the kernels that run once per frame."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc in this environment")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), os.path.join(ROOT, "dc_tts_amd", "csrc", "dctts_api.hip")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines():
        m = re.match(r"(\S.*?)\s+vgpr=(\d+) agpr=(\d+) sgpr=(\d+) scratch=(\d+) occ=(\d+) lds=(\d+)", line)
        if m:
            rows[m.group(1).strip()] = dict(vgpr=int(m.group(2)), agpr=int(m.group(3)), scratch=int(m.group(5)), lds=int(m.group(7)))
    assert rows, r.stdout[-2000:]
    return rows


def find(rows, prefix):
    hits = {k: v for k, v in rows.items() if k.startswith(prefix)}
    assert hits, f"{prefix}: not in the resource table ({sorted(rows)[:8]} ...)"
    return hits


@pytest.mark.parametrize("kernel", ["xchain_kernel<false, 4>", "xchain_kernel<false, 3>", "xtail_kernel<false, 4>", "xtail_kernel<false, 3>", "xgroup_kernel<false>", "dctts::xcone_kernel"])
def test_team_kernels_do_not_spill(table, kernel):
    for name, r in find(table, kernel).items():
        assert r["scratch"] == 0, f"{name}: {r['scratch']} bytes of scratch per lane"
        assert r["vgpr"] + r["agpr"] <= 256, f"{name}: {r}"            # 512 threads: two waves per SIMD
        assert r["lds"] <= 160 * 1024, f"{name}: {r}"


def test_the_throughput_kernels_do_not_spill(table):
    """hconv_kernel's instantiations (TextEnc / SSRN): capping the highway kernels at 128 registers spilled 424 bytes in round 5 and was reverted."""
    bad = {k: v for k, v in find(table, "hconv_kernel<").items() if v["scratch"] != 0}
    assert not bad, bad
