"""The C ABI library loads (no GPU needed) and exports every symbol include/fvsrn.h declares."""
import ctypes
import os
import re

import util
from fvsrn_amd import capi


def declared_symbols():
    text = open(os.path.join(util.ROOT, "include", "fvsrn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fvsrn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libfvsrn.so does not export " + n


def test_python_binding_covers_every_declared_symbol():
    bound = {n for n, _, _ in capi.SYMBOLS}
    assert bound == set(declared_symbols())


def test_no_gpu_calls_fail_loudly_without_a_device():
    import pytest
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    from fvsrn_amd import volnet_io
    vn = util.random_network()
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    rc = capi.lib().fvsrn_evaluate_points(net._h, 16, None, 1, 16, 0, None)
    assert rc == -6 and b"no HIP device" in capi.lib().fvsrn_last_error()


def test_product_package_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under fv-srn_amd/ may import, link or load it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(util.ROOT, "fv-srn_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".inc", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"srn_oracle|from oracle|import oracle|oracle/", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_built_kernels_keep_the_dot_hazard_distance():
    """The latent-grid sums start with an inline-assembly v_dot2_f32_f16 (srn_device.hpp dot2_from_zero); hipcc's hazard recognizer
    does not see it, so the three wait states gfx950 wants between a DOT instruction and a reader of another opcode are the source's
    job.  tools/check_dot_hazard.py disassembles the objects of the build and reports every reader that comes too early."""
    import glob
    import subprocess
    import sys
    objs = sorted(glob.glob(os.path.join(util.ROOT, "fv-srn_amd", "csrc", "build", "kernels_*.o")))
    if not objs:
        pytest.skip("no object files (the library was built elsewhere)")
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "check_dot_hazard.py")] + objs, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:]
