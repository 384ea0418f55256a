"""
ctypes binding of ``libfvsrn.so`` (include/fvsrn.h).

This is the thin Python host layer used by the tests, ``bench.py`` and the multi-GPU driver: it only
moves pointers.  All arithmetic happens in the HIP kernels behind the C ABI; there is NO CPU fallback:
if the library or a GPU is missing the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FVSRN_LIBRARY: developer override for A/B builds of the same ABI (tools/ablate.sh)
LIB_PATH = os.environ.get("FVSRN_LIBRARY") or os.path.join(_HERE, "libfvsrn.so")

ACTIVATIONS = {"ReLU": 0, "Sine": 1, "Snake": 2, "SnakeAlt": 3, "Sigmoid": 4, "None": 5}
OUTPUT_MODES = {"density": 0, "density:direct": 1, "rgbo": 2, "rgbo:direct": 3, "densitygrad": 4,
                "densitygrad:direct": 5, "densitygrad:cubic": 6, "densitycurvature": 7, "densitycurvature:direct": 8}
TF_NONE, TF_IDENTITY, TF_GAUSSIAN, TF_PIECEWISE, TF_TEXTURE = range(5)
TF_COLS = {TF_NONE: 0, TF_IDENTITY: 0, TF_GAUSSIAN: 6, TF_PIECEWISE: 5, TF_TEXTURE: 4}
BLEND_ALPHA, BLEND_BEER_LAMBERT = 0, 1
ORIENTATIONS = {"Xp": 0, "Xm": 1, "Yp": 2, "Ym": 3, "Zp": 4, "Zm": 5}


class FvsrnError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message if message else "fvsrn error %d" % code)
        self.code = code


class NetworkInfo(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("hidden_channels", C.c_int), ("num_fourier", C.c_int),
                ("has_direction", C.c_int), ("has_time", C.c_int), ("use_direction_in_fourier", C.c_int),
                ("output_mode", C.c_int), ("output_channels", C.c_int), ("activation", C.c_int),
                ("activation_param", C.c_float), ("grid_channels", C.c_int), ("grid_encoding", C.c_int),
                ("grid_res", C.c_int * 3), ("time_num", C.c_int), ("ensemble_num", C.c_int),
                ("num_parameters", C.c_int), ("max_warps_shared", C.c_int), ("max_warps_mixed", C.c_int),
                ("flops_per_sample", C.c_double), ("mfma_flops_per_sample", C.c_double),
                ("box_min", C.c_float * 3), ("box_size", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("cam_eye", C.c_float * 3), ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("fov_y_radians", C.c_float), ("stepsize", C.c_float), ("density_min", C.c_float),
                ("density_max", C.c_float), ("early_out", C.c_int), ("blend_mode", C.c_int), ("tf_kind", C.c_int),
                ("tf_scale_absorption", C.c_float), ("tf_scale_emission", C.c_float),
                ("tf_table", C.POINTER(C.c_float)), ("tf_rows", C.c_int),
                ("gradient_mode", C.c_int), ("finite_differences_stepsize", C.c_float),
                ("brdf_enable_magnitude_scaling", C.c_int), ("brdf_enable_phong", C.c_int),
                ("brdf_magnitude_scaling", C.c_float), ("brdf_ambient", C.c_float), ("brdf_specular", C.c_float),
                ("brdf_magnitude_center", C.c_float), ("brdf_magnitude_radius", C.c_float),
                ("brdf_specular_exponent", C.c_int), ("brdf_light_type", C.c_int), ("brdf_light", C.c_float * 3),
                ("tf_preintegration", C.c_int), ("adjoint_grid_stepsize", C.c_float), ("tf_gaussian_mode", C.c_int)]


# fvsrn_option (include/fvsrn.h): tuning / developer switches of a handle
OPTIONS = {"small_kernel": 0, "persistent": 1, "depth_segments": 2, "fourier_resync": 3, "unit_quota": 4, "tile_order": 5,
           "waves_per_block": 6, "max_blocks_per_cu": 7, "relu_clamp": 8, "keyframe_slots": 9, "working_grids": 10, "overlap_kernel": 11,
           "persistent_reserve": 12, "cell_table": 13}
ERR_WRONG_DEVICE = -8

GRADIENT_OFF_OR_DIRECT, GRADIENT_FINITE_DIFFERENCES, GRADIENT_ADJOINT_METHOD = 0, 1, 2
LIGHT_POINT, LIGHT_DIRECTIONAL = 0, 1
PREINTEGRATE_NONE, PREINTEGRATE_1D, PREINTEGRATE_2D = 0, 1, 2
TF_GAUSSIAN_PLAIN, TF_GAUSSIAN_SCALE_WITH_GRADIENT, TF_GAUSSIAN_ANALYTIC = 0, 1, 2


# every symbol include/fvsrn.h declares: (name, restype, argtypes)
_VP, _SZ, _I, _F, _D = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
_FP, _DP, _U16P = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint16)
SYMBOLS = [
    ("fvsrn_last_error", C.c_char_p, []),
    ("fvsrn_version", C.c_char_p, []),
    ("fvsrn_device_count", _I, []),
    ("fvsrn_probe_stream_concurrency", _I, [C.POINTER(_VP), _I, _I, _FP]),
    ("fvsrn_network_create_from_volnet", _I, [_VP, _SZ, C.POINTER(_VP)]),
    ("fvsrn_network_create", _I, [C.POINTER(_VP)]),
    ("fvsrn_network_destroy", None, [_VP]),
    ("fvsrn_network_set_input", _I, [_VP, _I, _I, _FP, _I, _I, _I]),
    ("fvsrn_network_set_output_mode", _I, [_VP, _I]),
    ("fvsrn_network_add_layer", _I, [_VP, _FP, _FP, _I, _I, _I, _F]),
    ("fvsrn_network_set_box", _I, [_VP, _FP, _FP]),
    ("fvsrn_network_set_latent_grid_layout", _I, [_VP, _I, _I, _I, _I, _I]),
    ("fvsrn_network_set_latent_grid", _I, [_VP, _I, _I, _FP, _I, _I, _I, _I, _I, _DP]),
    ("fvsrn_network_valid", _I, [_VP]),
    ("fvsrn_network_save_volnet", _I, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    ("fvsrn_network_set_time_and_ensemble", _I, [_VP, _F, _I]),
    ("fvsrn_network_prepare", _I, [_VP, _VP]),
    ("fvsrn_network_clear_gpu_resources", _I, [_VP]),
    ("fvsrn_network_get_info", _I, [_VP, C.POINTER(NetworkInfo)]),
    ("fvsrn_network_get_layer", _I, [_VP, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _FP, _U16P, _U16P]),
    ("fvsrn_network_get_fourier", _I, [_VP, _U16P, _I, C.POINTER(_I)]),
    ("fvsrn_evaluate_points", _I, [_VP, _VP, _VP, _SZ, _VP, _I, _VP]),
    ("fvsrn_evaluate_points_half", _I, [_VP, _VP, _VP, _SZ, _VP, _I, _VP]),
    ("fvsrn_evaluate_points_adjoint", _I, [_VP, _VP, _VP, _SZ, _VP, _F, _I, _VP]),
    ("fvsrn_scene_desc_size", _SZ, []),
    ("fvsrn_network_info_size", _SZ, []),
    ("fvsrn_scene_create", _I, [C.POINTER(SceneDesc), C.POINTER(_VP)]),
    ("fvsrn_scene_update", _I, [_VP, C.POINTER(SceneDesc)]),
    ("fvsrn_scene_destroy", None, [_VP]),
    ("fvsrn_camera_on_a_sphere", _I, [_I, _DP, _D, _D, _D, _FP, _FP, _FP]),
    ("fvsrn_render", _I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    ("fvsrn_scene_last_render_info", _I, [_VP, C.POINTER(_I)]),
    ("fvsrn_stripe_rows", _I, [_I, _I, _I, _I]),
    ("fvsrn_render_stripes", _I, [_VP, _VP, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    ("fvsrn_render_stripes_batch", _I, [C.POINTER(_VP), C.POINTER(_VP), _I, _VP, _I, _I, _I, _I, _I, _I, _FP, _FP, _VP, _VP, _I, _F, _VP]),
    ("fvsrn_extract_color", _I, [_VP, _I, _I, _I, _I, _F, _VP, _VP]),
    ("fvsrn_extract_color_rgba8", _I, [_VP, _I, _I, _I, _I, _F, _VP, _VP]),
    ("fvsrn_depth_range", _I, [_VP, _I, _I, _VP, _VP]),
    ("fvsrn_extract_color_ranged", _I, [_VP, _I, _I, _I, _I, _F, _VP, _VP, _VP, _VP]),
    ("fvsrn_generate_rays", _I, [_FP, _FP, _FP, _F, _I, _I, _VP, _VP, _VP]),
    ("fvsrn_scene_evaluate_tf", _I, [_VP, _VP, _VP, _SZ, _F, _F, _F, _VP, _VP]),
    ("fvsrn_network_kernel_name", _I, [_VP, _I, C.c_char_p, _SZ]),
    ("fvsrn_scene_last_kernel_name", _I, [_VP, C.c_char_p, _SZ]),
    ("fvsrn_debug_state", _I, [C.c_char_p, _SZ]),
    ("fvsrn_network_set_option", _I, [_VP, _I, _I]),
    ("fvsrn_network_get_option", _I, [_VP, _I, C.POINTER(_I)]),
    ("fvsrn_scene_set_option", _I, [_VP, _I, _I]),
    ("fvsrn_scene_get_option", _I, [_VP, _I, C.POINTER(_I)]),
    ("fvsrn_network_keyframe_stats", _I, [_VP, C.POINTER(C.c_ulonglong)]),
    ("fvsrn_network_cell_table_stats", _I, [_VP, C.POINTER(C.c_ulonglong)]),
    ("fvsrn_volume_create", _I, [_VP, _I, _I, _I, _I, _I, _FP, _FP, C.POINTER(_VP)]),
    ("fvsrn_volume_destroy", _I, [_VP]),
    ("fvsrn_volume_load_cvol", _I, [C.c_char_p, _I, C.POINTER(_VP)]),
    ("fvsrn_volume_save_cvol", _I, [C.c_char_p, C.c_char_p, _VP, _I, _I, _I, _I, _F, _F, _F]),
    ("fvsrn_volume_save_cvol_compressed", _I, [C.c_char_p, C.c_char_p, _VP, _I, _I, _I, _I, _F, _F, _F, _I]),
    ("fvsrn_cvol_write", _I, [C.c_char_p, C.POINTER(_F), _I, _VP, C.POINTER(_VP), _I]),
    ("fvsrn_volume_info", _I, [_VP, C.POINTER(_I), _FP, _FP]),
    ("fvsrn_volume_get_data", _I, [_VP, _FP, _SZ]),
    ("fvsrn_cvol_read", _I, [C.c_char_p, _FP, _VP, _VP]),
    ("fvsrn_volume_evaluate_points", _I, [_VP, _I, _I, _I, _VP, _SZ, _VP, _VP]),
    ("fvsrn_render_volume", _I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
]

_lib = None


def lib() -> C.CDLL:
    """Loads libfvsrn.so (built by ``__graft_entry__.build()`` / ``make -C fv-srn_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FvsrnError(-6, "libfvsrn.so not found at %s: build it first (python -c 'import __graft_entry__ as g; "
                                 "g.build()'); there is no CPU fallback" % LIB_PATH)
        # Share ONE HIP runtime with PyTorch: torch ships its own libamdhip64.so.7; if it is loaded first our
        # library binds to it by SONAME.  (Loaded the other way round the process would hold two HIP/HSA
        # runtimes and the second one finds no device.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            if os.environ.get("FVSRN_LIBRARY") and not hasattr(l, name):
                continue  # (an A/B library of an older build, tools/variant.sh: entries it lacks fail when they are called)
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.fvsrn_scene_desc_size() != C.sizeof(SceneDesc) or l.fvsrn_network_info_size() != C.sizeof(NetworkInfo):
            raise FvsrnError(-1, "capi.py struct mirrors do not match include/fvsrn.h of the loaded libfvsrn.so")
        _lib = l
    return _lib


def _check(code: int) -> None:
    if code != 0:
        raise FvsrnError(code, lib().fvsrn_last_error().decode("utf-8", "replace"))


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def device_count() -> int:
    return lib().fvsrn_device_count()


def probe_stream_concurrency(streams=6, microseconds: int = 2000) -> float:
    """How many streams run side by side (fvsrn_probe_stream_concurrency).  streams: a count (fresh streams: ~min(count, hardware queues of the
    process)) or a list of HIP stream handles (ints, e.g. torch.cuda.Stream.cuda_stream): the caller's own streams."""
    out = C.c_float(0.0)
    if isinstance(streams, int):
        _check(lib().fvsrn_probe_stream_concurrency(None, int(streams), int(microseconds), C.byref(out)))
    else:
        handles = (_VP * len(streams))(*[_VP(int(h)) for h in streams])
        _check(lib().fvsrn_probe_stream_concurrency(handles, len(streams), int(microseconds), C.byref(out)))
    return float(out.value)


def _torch_ptr(t, dtype_name: str, what: str) -> int:
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FvsrnError(-1, "%s must be a CUDA(HIP) tensor" % what)
    if str(t.dtype) != dtype_name or not t.is_contiguous():
        raise FvsrnError(-1, "%s must be contiguous %s" % (what, dtype_name))
    return t.data_ptr()


def _current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class Network:
    """SceneNetwork handle (reference pyrenderer.SceneNetwork, volume_interpolation_network.cpp:1928-1972)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    @staticmethod
    def from_volnet(data: bytes) -> "Network":
        h = _VP()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        _check(lib().fvsrn_network_create_from_volnet(C.cast(buf, _VP), len(data), C.byref(h)))
        return Network(h.value)

    @staticmethod
    def create() -> "Network":
        h = _VP()
        _check(lib().fvsrn_network_create(C.byref(h)))
        return Network(h.value)

    def __del__(self):
        try:
            if self._h:
                lib().fvsrn_network_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- builder (mirrors export_to_pyrenderer's calls) -------------------------------------------------
    def set_input(self, fourier: Optional[np.ndarray], has_time=False, has_direction=False, premultiplied=True):
        if fourier is None:
            _check(lib().fvsrn_network_set_input(self._h, int(has_time), int(has_direction), None, 0, 3, 1))
        else:
            f = np.ascontiguousarray(fourier, dtype=np.float32)
            _check(lib().fvsrn_network_set_input(self._h, int(has_time), int(has_direction), _fptr(f), f.shape[0],
                                                 f.shape[1], int(premultiplied)))

    def set_output_mode(self, mode: str):
        _check(lib().fvsrn_network_set_output_mode(self._h, OUTPUT_MODES[mode]))

    def add_layer(self, weight: np.ndarray, bias: np.ndarray, activation: str, param: float = 1.0):
        w = np.ascontiguousarray(weight, dtype=np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32)
        _check(lib().fvsrn_network_add_layer(self._h, _fptr(w), _fptr(b), w.shape[0], w.shape[1],
                                             ACTIVATIONS[activation], float(param)))

    def set_box(self, box_min: Sequence[float], box_size: Sequence[float]):
        a = np.asarray(box_min, np.float32)
        b = np.asarray(box_size, np.float32)
        _check(lib().fvsrn_network_set_box(self._h, _fptr(a), _fptr(b)))

    def set_latent_grid_layout(self, time_min, time_num, time_step, ensemble_min, ensemble_num):
        _check(lib().fvsrn_network_set_latent_grid_layout(self._h, time_min, time_num, time_step, ensemble_min, ensemble_num))

    def set_latent_grid(self, is_ensemble: bool, index: int, grid: np.ndarray, encoding: int) -> float:
        g = np.ascontiguousarray(grid, dtype=np.float32)
        if g.ndim == 5:
            g = g[0]
        err = C.c_double(0)
        _check(lib().fvsrn_network_set_latent_grid(self._h, int(is_ensemble), index, _fptr(g), g.shape[0], g.shape[1],
                                                   g.shape[2], g.shape[3], encoding, C.byref(err)))
        return err.value

    # -- queries -----------------------------------------------------------------------------------------
    def valid(self) -> bool:
        return lib().fvsrn_network_valid(self._h) == 1

    def last_error(self) -> str:
        return lib().fvsrn_last_error().decode()

    def save(self) -> bytes:
        n = _SZ(0)
        _check(lib().fvsrn_network_save_volnet(self._h, None, 0, C.byref(n)))
        buf = (C.c_char * n.value)()
        _check(lib().fvsrn_network_save_volnet(self._h, C.cast(buf, _VP), n.value, C.byref(n)))
        return bytes(buf)

    def info(self) -> NetworkInfo:
        i = NetworkInfo()
        _check(lib().fvsrn_network_get_info(self._h, C.byref(i)))
        return i

    def layer(self, index: int):
        co, ci, act, p = _I(), _I(), _I(), _F()
        _check(lib().fvsrn_network_get_layer(self._h, index, C.byref(co), C.byref(ci), C.byref(act), C.byref(p), None, None))
        w = np.zeros(co.value * ci.value, np.uint16)
        b = np.zeros(co.value, np.uint16)
        _check(lib().fvsrn_network_get_layer(self._h, index, None, None, None, None, w.ctypes.data_as(_U16P),
                                             b.ctypes.data_as(_U16P)))
        return co.value, ci.value, act.value, p.value, w, b

    def fourier(self) -> np.ndarray:
        n = _I()
        _check(lib().fvsrn_network_get_fourier(self._h, None, 0, C.byref(n)))
        m = np.zeros(n.value, np.uint16)
        _check(lib().fvsrn_network_get_fourier(self._h, m.ctypes.data_as(_U16P), n.value, C.byref(n)))
        return m

    def kernel_name(self, render: bool = True) -> str:
        buf = C.create_string_buffer(256)
        _check(lib().fvsrn_network_kernel_name(self._h, int(render), buf, 256))
        return buf.value.decode()

    def set_time_and_ensemble(self, time: float, ensemble: int = 0):
        _check(lib().fvsrn_network_set_time_and_ensemble(self._h, float(time), int(ensemble)))

    def prepare(self, stream: Optional[int] = None):
        """Uploads (first use) and the key-frame blend of the current time on `stream`, ahead of the next launch (fvsrn_network_prepare)."""
        _check(lib().fvsrn_network_prepare(self._h, _current_stream() if stream is None else stream))

    def set_option(self, name: str, value: int):
        _check(lib().fvsrn_network_set_option(self._h, OPTIONS[name], int(value)))
        self._info = None

    def get_option(self, name: str) -> int:
        v = _I()
        _check(lib().fvsrn_network_get_option(self._h, OPTIONS[name], C.byref(v)))
        return v.value

    def keyframe_stats(self) -> dict:
        a = (C.c_ulonglong * 6)()
        _check(lib().fvsrn_network_keyframe_stats(self._h, a))
        return dict(zip(("key_frames", "slots", "uploads", "on_demand", "prefetched", "bytes"), [int(v) for v in a]))

    def cell_table_stats(self) -> dict:
        a = (C.c_ulonglong * 4)()
        _check(lib().fvsrn_network_cell_table_stats(self._h, a))
        return dict(zip(("table_bytes", "builds", "builds_plain", "resident_bytes"), [int(v) for v in a]))

    def clear_gpu_resources(self):
        _check(lib().fvsrn_network_clear_gpu_resources(self._h))

    # -- IVolumeInterpolation.evaluate (volume_interpolation.cpp:26-127) ---------------------------------
    def evaluate(self, positions, directions=None, out=None, stream: Optional[int] = None, world: bool = False, predicted_gradient: bool = False):
        """positions: unit-box coordinates like the reference; world=True maps world positions through the box.  The dtype of `positions`
        picks the entry point like the reference's scalar-type dispatch: float32 -> fvsrn_evaluate_points, float16 (positions, directions
        and values all fp16, 8 bytes per point of a scalar network) -> fvsrn_evaluate_points_half.  predicted_gradient: (n,4) = value +
        the gradient a densitygrad* / densitycurvature* network predicts (FVSRN_EVAL_WITH_PREDICTED_GRADIENT)."""
        import torch
        n = positions.shape[0]
        half = positions.dtype == torch.float16
        dt = "torch.float16" if half else "torch.float32"
        pp = _torch_ptr(positions, dt, "positions")
        dp = _torch_ptr(directions, dt, "directions") if directions is not None else None
        oc = 4 if predicted_gradient else self.info().output_channels
        if out is None:
            out = torch.empty((n, oc), dtype=positions.dtype, device=positions.device)
        op = _torch_ptr(out, dt, "out")
        fn = lib().fvsrn_evaluate_points_half if half else lib().fvsrn_evaluate_points
        _check(fn(self._h, pp, dp, n, op, (1 if world else 0) | (2 if predicted_gradient else 0), _current_stream() if stream is None else stream))
        return out

    def evaluate_with_gradients_and_curvature(self, positions, directions=None, stream: Optional[int] = None, world: bool = False):
        """IVolumeInterpolation.evaluate_with_gradients_and_curvature of a network that predicts all three (output mode
        densitycurvature[:direct], volume_interpolation.cpp:245-360): (n,1) values, (n,3) gradients, (n,2) curvature values."""
        import torch
        n = positions.shape[0]
        pp = _torch_ptr(positions, "torch.float32", "positions")
        dp = _torch_ptr(directions, "torch.float32", "directions") if directions is not None else None
        out = torch.empty((n, 6), dtype=torch.float32, device=positions.device)
        _check(lib().fvsrn_evaluate_points(self._h, pp, dp, n, _torch_ptr(out, "torch.float32", "out"), (1 if world else 0) | 4,
                                           _current_stream() if stream is None else stream))
        return out[:, 0:1].contiguous(), out[:, 1:4].contiguous(), out[:, 4:6].contiguous()

    def evaluate_with_adjoint_gradient(self, positions, directions=None, grid_step: float = 0.0, stream: Optional[int] = None,
                                       world: bool = False):
        """IVolumeInterpolation.evaluate_with_gradients in GRADIENT_MODE_ADJOINT_METHOD (volume_interpolation.cpp:128-243): (n,1) values
        and (n,3) analytic gradients w.r.t. the normalized position; grid_step = 0: 1 / (4 * latent grid resolution)."""
        import torch
        n = positions.shape[0]
        pp = _torch_ptr(positions, "torch.float32", "positions")
        dp = _torch_ptr(directions, "torch.float32", "directions") if directions is not None else None
        out = torch.empty((n, 4), dtype=torch.float32, device=positions.device)
        _check(lib().fvsrn_evaluate_points_adjoint(self._h, pp, dp, n, _torch_ptr(out, "torch.float32", "out"), float(grid_step),
                                                   1 if world else 0, _current_stream() if stream is None else stream))
        return out[:, 0:1].contiguous(), out[:, 1:4].contiguous()


class Scene:
    """Camera + DVR + TF + blending POD (reference: the module tree below ImageEvaluatorSimple)."""

    def __init__(self, **kw):
        self._h = None
        self._keep = None
        d = self._desc(**kw)
        h = _VP()
        _check(lib().fvsrn_scene_create(C.byref(d), C.byref(h)))
        self._h = C.c_void_p(h.value)

    def _desc(self, *, eye, right, up, fov_y_radians, stepsize, density_min=0.0, density_max=1.0, early_out=True,
              blend_mode=BLEND_BEER_LAMBERT, tf_kind=TF_IDENTITY, tf_scale_absorption=1.0, tf_scale_emission=1.0,
              tf_table=None, gradient_mode=GRADIENT_OFF_OR_DIRECT, finite_differences_stepsize=0.0, brdf=None,
              tf_preintegration=PREINTEGRATE_NONE, adjoint_grid_stepsize=0.0, tf_gaussian_mode=TF_GAUSSIAN_PLAIN) -> SceneDesc:
        d = SceneDesc()
        d.cam_eye[:] = [float(v) for v in eye]
        d.cam_right[:] = [float(v) for v in right]
        d.cam_up[:] = [float(v) for v in up]
        d.fov_y_radians = fov_y_radians
        d.stepsize = stepsize
        d.density_min, d.density_max = density_min, density_max
        d.early_out = int(early_out)
        d.blend_mode = blend_mode
        d.tf_kind = tf_kind
        d.tf_scale_absorption, d.tf_scale_emission = tf_scale_absorption, tf_scale_emission
        if tf_table is not None:
            t = np.ascontiguousarray(tf_table, dtype=np.float32)
            assert t.ndim == 2 and t.shape[1] == TF_COLS[tf_kind], "TF table shape does not match its kind"
            self._keep = t
            d.tf_table = _fptr(t)
            d.tf_rows = t.shape[0]
        else:
            d.tf_table = None
            d.tf_rows = 0
        d.gradient_mode = gradient_mode
        d.finite_differences_stepsize = finite_differences_stepsize
        d.tf_preintegration = tf_preintegration
        d.adjoint_grid_stepsize = adjoint_grid_stepsize
        d.tf_gaussian_mode = tf_gaussian_mode
        if brdf:  # BRDFLambert: dict(enable_phong=, enable_magnitude_scaling=, magnitude_scaling=, ambient=, specular=,
            #                     magnitude_center=, magnitude_radius=, specular_exponent=, light_type=, light=(x,y,z))
            d.brdf_enable_phong = int(brdf.get("enable_phong", False))
            d.brdf_enable_magnitude_scaling = int(brdf.get("enable_magnitude_scaling", False))
            d.brdf_magnitude_scaling = brdf.get("magnitude_scaling", 1.0)
            d.brdf_ambient = brdf.get("ambient", 0.1)
            d.brdf_specular = brdf.get("specular", 0.1)
            d.brdf_magnitude_center = brdf.get("magnitude_center", 0.5)
            d.brdf_magnitude_radius = brdf.get("magnitude_radius", 0.1)
            d.brdf_specular_exponent = int(brdf.get("specular_exponent", 16))
            d.brdf_light_type = int(brdf.get("light_type", LIGHT_POINT))
            d.brdf_light[:] = [float(v) for v in brdf.get("light", (0.0, 0.0, 1.0))]
        return d

    def evaluate_tf(self, densities, density_min: float, density_max: float, previous=None, stepsize: float = 1.0, stream=None):
        """ITransferFunction::evaluate / evaluate_with_previous on a (B,1) fp32 CUDA tensor -> (B,4)."""
        import torch
        n = int(densities.shape[0])
        out = torch.empty((n, 4), dtype=torch.float32, device=densities.device)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(lib().fvsrn_scene_evaluate_tf(self._h, _torch_ptr(densities, "torch.float32", "densities"),
                                             _torch_ptr(previous, "torch.float32", "previous densities") if previous is not None else None,
                                             n, density_min, density_max, stepsize, out.data_ptr(), s))
        return out

    def update(self, **kw):
        d = self._desc(**kw)
        _check(lib().fvsrn_scene_update(self._h, C.byref(d)))

    def set_option(self, name: str, value: int):
        _check(lib().fvsrn_scene_set_option(self._h, OPTIONS[name], int(value)))
        return self

    def get_option(self, name: str) -> int:
        v = _I()
        _check(lib().fvsrn_scene_get_option(self._h, OPTIONS[name], C.byref(v)))
        return v.value

    def last_render_info(self) -> dict:
        """How the last render of this scene treated a ray's samples (fvsrn_scene_last_render_info)."""
        a = (_I * 4)()
        _check(lib().fvsrn_scene_last_render_info(self._h, a))
        return dict(segments=a[0], rotation_resync=a[1], resident_kernel=a[2] in (1, 4), overlap_kernel=a[2] == 2, adjoint_kernel=a[2] == 3, cell_table=a[2] in (4, 5),
                    waves_per_block=a[3])

    def last_kernel_name(self) -> str:
        """The kernel the last render of this scene launched (fvsrn_scene_last_kernel_name)."""
        buf = C.create_string_buffer(256)
        _check(lib().fvsrn_scene_last_kernel_name(self._h, buf, 256))
        return buf.value.decode()

    def __del__(self):
        try:
            if self._h:
                lib().fvsrn_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def render(self, net: Network, width: int, height: int, y0: int = 0, y1: Optional[int] = None, out=None,
               stats=None, stream: Optional[int] = None):
        """ImageEvaluatorSimple::render: returns the (1,8,H,W) fp32 tensor (rows [y0,y1) written)."""
        import torch
        if y1 is None:
            y1 = height
        if out is None:
            out = torch.zeros((1, 8, height, width), dtype=torch.float32, device="cuda")
        op = _torch_ptr(out, "torch.float32", "out")
        sp = _torch_ptr(stats, "torch.int64", "stats") if stats is not None else None
        _check(lib().fvsrn_render(self._h, net._h, width, height, y0, y1, op, sp,
                                  _current_stream() if stream is None else stream))
        return out


def debug_state() -> str:
    """fvsrn_debug_state: the last launch of every live scene (kernel, launch shape, busy stream, device work counters); never blocks."""
    buf = C.create_string_buffer(16384)
    _check(lib().fvsrn_debug_state(buf, 16384))
    return buf.value.decode(errors="replace")


def stripe_rows(height: int, stripe: int, rank: int, world: int) -> int:
    return lib().fvsrn_stripe_rows(height, stripe, rank, world)


VOLUME_U8, VOLUME_U16, VOLUME_F32 = range(3)
VOLUME_NEAREST, VOLUME_TRILINEAR, VOLUME_TRICUBIC = range(3)
VOLUME_SOURCE_TEXTURE, VOLUME_SOURCE_TENSOR = range(2)
_VOLUME_DTYPES = {"uint8": VOLUME_U8, "uint16": VOLUME_U16, "float32": VOLUME_F32}


class Volume:
    """One scalar feature of a grid volume (renderer/volume.h) for ``VolumeInterpolationGrid``: fvsrn_volume_* of the C ABI."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_array(cls, data_xyz, box_min=None, box_size=None) -> "Volume":
        """(X,Y,Z) uint8 / uint16 / float32 array.  Default box: [-s/2, s/2] with s = res / max(res), like
        VolumeInterpolationGrid::setSource(tensor) (volume_interpolation_grid.cpp:200-224)."""
        a = np.ascontiguousarray(data_xyz)
        if a.ndim != 3 or a.dtype.name not in _VOLUME_DTYPES:
            raise ValueError("expected a (X,Y,Z) uint8 / uint16 / float32 array")
        if box_size is None:
            box_size = [n / max(a.shape) for n in a.shape]
        if box_min is None:
            box_min = [-0.5 * b for b in box_size]
        h = _VP()
        _check(lib().fvsrn_volume_create(a.ctypes.data_as(_VP), _VOLUME_DTYPES[a.dtype.name], a.shape[0], a.shape[1], a.shape[2], 0,
                                         (C.c_float * 3)(*[float(v) for v in box_min]), (C.c_float * 3)(*[float(v) for v in box_size]),
                                         C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path: str, feature_index: int = 0) -> "Volume":
        """.cvol, version 1 or the old density-only format, uncompressed or LZ4 (Volume::Volume(filename), volume.cpp:685-793); box = [-world/2, world/2]."""
        h = _VP()
        _check(lib().fvsrn_volume_load_cvol(os.fsencode(path), feature_index, C.byref(h)))
        return cls(h)

    @staticmethod
    def save_cvol(path: str, data_xyz, world_size=(1.0, 1.0, 1.0), feature_name: str = "density", compression: int = 0) -> None:
        """Volume::save(filename, compression) (volume.cpp:623-682): compression 0 .. 9, > 0 = LZ4 messages (Flag_Compressed)"""
        a = np.asarray(data_xyz)
        if a.ndim != 3 or a.dtype.name not in _VOLUME_DTYPES:
            raise ValueError("expected a (X,Y,Z) uint8 / uint16 / float32 array")
        xfast = np.ascontiguousarray(a.transpose(2, 1, 0))  # file order: x fastest, z slowest
        _check(lib().fvsrn_volume_save_cvol_compressed(os.fsencode(path), feature_name.encode(), xfast.ctypes.data_as(_VP), _VOLUME_DTYPES[a.dtype.name],
                                                       a.shape[0], a.shape[1], a.shape[2], *[float(w) for w in world_size], int(compression)))

    def info(self):
        res, bmin, bsize = (_I * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
        _check(lib().fvsrn_volume_info(self._h, res, bmin, bsize))
        return tuple(res), np.array(bmin, np.float32), np.array(bsize, np.float32)

    def data(self) -> np.ndarray:
        """(X,Y,Z) float32 copy of the voxels (u8 / u16 volumes: normalised to [0,1])."""
        res = self.info()[0]
        a = np.empty(res[0] * res[1] * res[2], np.float32)
        _check(lib().fvsrn_volume_get_data(self._h, _fptr(a), a.size))
        return a.reshape(res[2], res[1], res[0]).transpose(2, 1, 0)

    def evaluate(self, positions, interpolation=VOLUME_TRILINEAR, source=VOLUME_SOURCE_TEXTURE, new_behavior=False, stream=None):
        """IVolumeInterpolation::evaluate: (N,3) fp32 CUDA world positions -> (N,1)."""
        import torch
        pp = _torch_ptr(positions, "torch.float32", "positions")
        out = torch.empty((positions.shape[0], 1), dtype=torch.float32, device=positions.device)
        _check(lib().fvsrn_volume_evaluate_points(self._h, source, interpolation, int(new_behavior), pp, positions.shape[0],
                                                  _torch_ptr(out, "torch.float32", "out"), _current_stream() if stream is None else stream))
        return out

    def render(self, scene: "Scene", width: int, height: int, interpolation=VOLUME_TRILINEAR, source=VOLUME_SOURCE_TEXTURE,
               new_behavior=False, out=None, stats=None, stream=None, provide_normals=False):
        """ImageEvaluatorSimple.render with this volume: (1,8,H,W) fp32."""
        import torch
        if out is None:
            out = torch.zeros((1, 8, height, width), dtype=torch.float32, device="cuda")
        sp = _torch_ptr(stats, "torch.int64", "stats") if stats is not None else None
        _check(lib().fvsrn_render_volume(scene._h, self._h, source, interpolation, int(new_behavior), int(provide_normals), width, height,
                                         _torch_ptr(out, "torch.float32", "out"), sp, _current_stream() if stream is None else stream))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().fvsrn_volume_destroy(self._h)
                self._h = None
        except Exception:
            pass


def render_stripes(scene: Scene, net: Network, width: int, height: int, stripe: int, rank: int, world: int, out=None,
                   stats=None, stream: Optional[int] = None):
    """Renders the rows {y : (y // stripe) % world == rank} into a compact (8, local_rows, W) tensor."""
    import torch
    rows = stripe_rows(height, stripe, rank, world)
    if out is None:
        out = torch.zeros((8, rows, width), dtype=torch.float32, device="cuda")
    op = _torch_ptr(out, "torch.float32", "out")
    sp = _torch_ptr(stats, "torch.int64", "stats") if stats is not None else None
    _check(lib().fvsrn_render_stripes(scene._h, net._h, width, height, stripe, rank, world, op, sp,
                                      _current_stream() if stream is None else stream))
    return out


def render_stripes_batch(scenes, streams, net: Network, width: int, height: int, stripe: int, rank: int, world: int, cameras, times=None,
                         out=None, rgba8=None, use_tonemapping: bool = False, max_exposure: float = 1.0, stats=None):
    """fvsrn_render_stripes_batch: `len(cameras)` frames in ONE call.  scenes / streams: one lane or several (frame f runs on lane f % lanes:
    scenes[lane] on the raw stream handle streams[lane]); cameras: (frames, 9) = eye, right, up per frame; times: optional per-frame time.
    out: (frames, 8, rows, W) fp32 (allocated when None; world == 1: rows = H); rgba8: optional (frames, rows, W) int32 that also receives every frame
    as packed RGBA8 words (COLOR mode).  Returns out."""
    import torch
    cams = np.ascontiguousarray(cameras, np.float32).reshape(-1, 9)
    n = int(cams.shape[0])
    rows = height if world == 1 else stripe_rows(height, stripe, rank, world)
    if out is None:
        out = torch.zeros((n, 8, rows, width), dtype=torch.float32, device="cuda")
    if out.numel() < n * 8 * rows * width:
        raise FvsrnError(-1, "out must hold (frames, 8, rows, W) floats")
    if rgba8 is not None and rgba8.numel() < n * rows * width:
        raise FvsrnError(-1, "rgba8 must hold (frames, rows, W) words")
    lanes = len(scenes)
    if lanes != len(streams) or lanes < 1:
        raise FvsrnError(-1, "one stream per scene")
    hs = (_VP * lanes)(*[sc._h for sc in scenes])
    st = (_VP * lanes)(*[int(x) for x in streams])
    tp = None
    if times is not None:
        tarr = np.ascontiguousarray(times, np.float32).reshape(-1)
        if tarr.size != n:
            raise FvsrnError(-1, "one time per frame")
        tp = tarr.ctypes.data_as(_FP)
    _check(lib().fvsrn_render_stripes_batch(hs, st, lanes, net._h, width, height, stripe, rank, world, n, cams.ctypes.data_as(_FP), tp,
                                            _torch_ptr(out, "torch.float32", "out"), _torch_ptr(rgba8, "torch.int32", "rgba8") if rgba8 is not None else None,
                                            int(use_tonemapping), max_exposure, _torch_ptr(stats, "torch.int64", "stats") if stats is not None else None))
    return out


def depth_range(raw, stream=None):
    """fvsrn_depth_range of a (1,8,H,W) / (8,H,W) fp32 CUDA image (part): (3,) fp32 = {-min, max, nan flag} of the depth plane, mergeable over the parts
    of an image by an element-wise maximum (one all-reduce)."""
    import torch
    H, W = int(raw.shape[-2]), int(raw.shape[-1])
    out = torch.empty(3, dtype=torch.float32, device=raw.device)
    _check(lib().fvsrn_depth_range(_torch_ptr(raw, "torch.float32", "raw input"), W, H, out.data_ptr(), torch.cuda.current_stream().cuda_stream if stream is None else stream))
    return out


def extract_color_part(raw, channel_mode: int = 3, use_tonemapping: bool = False, max_exposure: float = 1.0, rgba8: bool = True, depth_range3=None, out=None,
                       stream=None):
    """fvsrn_extract_color_ranged on a part of an image -- a (8, rows, W) compact stripe image of one rank: (rows, W) int32 RGBA8 words, or (4, rows, W)
    fp32.  depth_range3: the merged depth range of the whole frame (CHANNEL_DEPTH)."""
    import torch
    if raw.dim() != 3 or raw.shape[0] != 8:
        raise FvsrnError(-1, "raw input must be of shape (8,rows,W)")
    H, W = int(raw.shape[1]), int(raw.shape[2])
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rp = depth_range3.data_ptr() if depth_range3 is not None else None
    if out is None:
        out = torch.empty((H, W), dtype=torch.int32, device=raw.device) if rgba8 else torch.empty((4, H, W), dtype=torch.float32, device=raw.device)
    _check(lib().fvsrn_extract_color_ranged(_torch_ptr(raw, "torch.float32", "raw input"), W, H, channel_mode, int(use_tonemapping), max_exposure, rp,
                                            None if rgba8 else out.data_ptr(), out.data_ptr() if rgba8 else None, s))
    return out


def camera_on_a_sphere(orientation: str, center, pitch: float, yaw: float, distance: float):
    """CameraOnASphere -> (eye, right, up) fp32 (renderer/camera.cpp:458-490,553-581)."""
    c = (C.c_double * 3)(*[float(v) for v in center])
    eye, right, up = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    _check(lib().fvsrn_camera_on_a_sphere(ORIENTATIONS[orientation], c, pitch, yaw, distance, eye, right, up))
    return np.array(eye, np.float32), np.array(right, np.float32), np.array(up, np.float32)


CHANNEL_MASK, CHANNEL_NORMAL, CHANNEL_DEPTH, CHANNEL_COLOR = range(4)


def extract_color(raw, channel_mode: int = CHANNEL_COLOR, use_tonemapping: bool = False, max_exposure: float = 1.0,
                  rgba8: bool = False, stream=None):
    """IImageEvaluator::ExtractColor on a (1,8,H,W) fp32 CUDA tensor: (1,4,H,W) fp32, or (H,W) int32 words 0xAABBGGRR."""
    import torch
    if raw.dim() != 4 or raw.shape[0] != 1 or raw.shape[1] != 8:
        raise FvsrnError(-1, "raw input must be of shape (1,8,H,W)")
    H, W = int(raw.shape[2]), int(raw.shape[3])
    src = _torch_ptr(raw, "torch.float32", "raw input")
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    if rgba8:
        out = torch.empty((H, W), dtype=torch.int32, device=raw.device)
        _check(lib().fvsrn_extract_color_rgba8(src, W, H, channel_mode, int(use_tonemapping), max_exposure, out.data_ptr(), s))
    else:
        out = torch.empty((1, 4, H, W), dtype=torch.float32, device=raw.device)
        _check(lib().fvsrn_extract_color(src, W, H, channel_mode, int(use_tonemapping), max_exposure, out.data_ptr(), s))
    return out


def generate_rays(eye, right, up, fov_y_radians: float, width: int, height: int, stream=None):
    """ICamera::generateRays: (ray_start, ray_dir), each (1,H,W,3) fp32 on the GPU."""
    import torch
    start = torch.empty((1, height, width, 3), dtype=torch.float32, device="cuda")
    direction = torch.empty_like(start)
    e, r, u = (np.ascontiguousarray(v, np.float32) for v in (eye, right, up))
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    _check(lib().fvsrn_generate_rays(_fptr(e), _fptr(r), _fptr(u), fov_y_radians, width, height, start.data_ptr(), direction.data_ptr(), s))
    return start, direction
