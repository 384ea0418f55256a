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
# NB (r04): importing this package no longer sets GPU_MAX_HW_QUEUES in os.environ -- a library import must not change the process environment,
# and the variable only counts if it is set before the HIP runtime starts.  bench.py and tiles.launch_ranks() export it for the processes they
# start; tiles.StripeRenderer measures what the process got and warns (tiles.py, "Hardware queues").
from . import capi, volnet_io  # noqa: F401

__all__ = ["capi", "volnet_io"]
