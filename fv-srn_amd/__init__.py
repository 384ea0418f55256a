"""
fv-srn_amd: MI355X-native (gfx950) implementation of ONE hot path of shamanDevel/fV-SRN --
the fused SRN-MLP + DVR ray-stepping inference renderer.

Layout
  csrc/          HIP kernels + C++ host model + the C ABI (include/fvsrn.h)  -> libfvsrn.so
  capi.py        ctypes binding of the C ABI (pointers only)
  volnet_io.py   pure-Python .volnet writer/reader (export script side, format cross-check)
  tiles.py       multi-GPU row-stripe partition + RCCL gather (torch.distributed)
  pyrenderer/    C++ pybind11 module mirroring the reference's `pyrenderer` surface

The directory name contains a hyphen (it is the name the project layout prescribes); import it with
    import importlib; fvsrn = importlib.import_module("fv-srn_amd")
or use the `fvsrn_amd` alias module at the repository root.
"""
import os

# The frame pipeline of a rank (tiles.StripeRenderer) runs on five streams -- two render streams, the collective's, the library's copy
# stream for key frames, the caller's -- and ROCm maps all streams of a process onto FOUR hardware queues by default: two streams that
# share a queue run in submission order, and the collective of frame i ended up behind the render of frame i + 1 (r03: a rank's share
# at world 8 at 80 % of frame / world instead of 98 %, profiles/r03/stripe_pipeline_r03.md).  The setting is read when the HIP runtime
# starts, so this only helps when the package is imported before the first HIP call of the process (torch.cuda.*); bench.py and
# tiles.launch_ranks() set it in the environment of the processes they start.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import capi, volnet_io  # noqa: E402,F401

__all__ = ["capi", "volnet_io"]
