"""The compiled tile kernels hold no vector read of a matrix instruction's result inside the wait states the hardware needs, on
any path (tools/check_mfma_hazards.py: the compiler pads the fall-through side of a branch only).  Static: the
sources are compiled to gfx950 assembly with the flags the shipped objects are built with; no GPU."""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["letkf_tile2.hip", "letkf_tile2f.hip", "letkf_tile2p.hip", "letkf_tile2w.hip", "lketkf_tile.hip", "letkf_tile_split.hip"]


def _check(name):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"),
                          os.path.join(ROOT, "torch-assimilate_amd", "csrc", name)], capture_output=True, text=True)
    return name, res.returncode, (res.stdout + res.stderr)[-2000:]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_tile_kernels_have_no_unpadded_mfma_reads():
    with ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(_check, SOURCES))
    bad = [(n, out) for n, rc, out in results if rc != 0]
    assert not bad, "\n".join("%s:\n%s" % b for b in bad)
