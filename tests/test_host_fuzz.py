"""Mutation fuzz of the .volnet / .cvol / scene-JSON parsers behind the C ABI (tools/host_fuzz.py), a short run on the normal
build; tools/run_asan.sh runs thousands of mutations on the ASan + UBSan build (profiles/r02/asan_report.txt)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mutated_inputs_are_rejected_with_messages_never_crash():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_fuzz.py"), "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "volnet: 160 mutated inputs" in r.stdout and "0 crashes" in r.stdout
