"""The committed fixtures are what the committed generator produces: `make_golden.py --check` imports the reference's Python
from /root/reference, regenerates every fixture into a temporary directory and compares array by array.  Skipped where the
reference checkout does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/applications/volnet"), reason="needs the reference checkout")
def test_fixtures_are_reproduced_bit_for_bit_by_the_generator():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "reproduce bit for bit" in r.stdout


@pytest.mark.skipif(not os.path.exists("/root/reference/applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol"), reason="needs the reference checkout")
def test_cvol_fixture_is_reproduced_by_its_generator():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_cvol_fixture.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "reproduces" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
