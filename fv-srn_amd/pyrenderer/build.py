#!/usr/bin/env python3
"""
Builds the `pyrenderer` pybind11 module in-tree (fv-srn_amd/pyrenderer/pyrenderer*.so).

Plain host C++ (g++) against the torch headers for the tensor type casters; it links libfvsrn.so (the C ABI)
and carries an $ORIGIN-relative rpath so the pair can be moved together.  No GPU needed to build.
"""
import glob
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> None:
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(HERE, "pyrenderer.cpp")
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "pyrenderer" + ext)
    lib = os.path.join(HERE, "..", "libfvsrn.so")
    if not os.path.exists(lib):
        raise SystemExit("build libfvsrn.so first (make -C fv-srn_amd/csrc)")
    deps = [src, os.path.join(HERE, "..", "..", "include", "fvsrn.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-attributes",
           "-DTORCH_EXTENSION_NAME=pyrenderer", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in cpp_extension.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-isystem", sysconfig.get_paths()["include"], src, "-o", out,
            "-L" + os.path.join(HERE, ".."), "-lfvsrn", "-L" + torch_lib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
            "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + torch_lib]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


if __name__ == "__main__":
    main()
