"""
ctypes front end of the CPU oracle (oracle/srn_oracle.c).        *** TEST INFRASTRUCTURE ONLY ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

``OracleNetwork`` assembles, from the layers of a .volnet (``fv-srn_amd/volnet_io.VolnetData``), the
arrays of the reference's constant block exactly the way ``SceneNetwork::fillConstantMemory`` does
(renderer/volume_interpolation_network.cpp:1236-1417), and hands them to the C restatement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FVSRN_ORACLE_LIBRARY") or os.path.join(_HERE, "libsrn_oracle.so")  # override: the sanitizer build (tools/run_asan.sh)

ACC_HALF, ACC_FLOAT, ACC_DEVICE, ACC_EXACT = 0, 1, 2, 3
ACTIVATIONS = {"ReLU": 0, "Sine": 1, "Snake": 2, "SnakeAlt": 3, "Sigmoid": 4}
OUTPUT_MODES = {"density": 0, "density:direct": 1, "rgbo": 2, "rgbo:direct": 3, "densitygrad": 4,
                "densitygrad:direct": 5, "densitygrad:cubic": 6, "densitycurvature": 7, "densitycurvature:direct": 8}
TF_NONE, TF_IDENTITY, TF_GAUSSIAN, TF_PIECEWISE, TF_TEXTURE = range(5)
BLEND_ALPHA, BLEND_BEER_LAMBERT = 0, 1

_U16P = C.POINTER(C.c_uint16)
_FP = C.POINTER(C.c_float)


class _OracleNet(C.Structure):
    _fields_ = [("C", C.c_int), ("F", C.c_int), ("G", C.c_int), ("NH", C.c_int), ("Cout", C.c_int),
                ("outputMode", C.c_int), ("activation", C.c_int), ("gridEncoding", C.c_int), ("passTime", C.c_int),
                ("accMode", C.c_int), ("useDirection", C.c_int), ("actParam", C.c_float), ("boxMin", C.c_float * 3), ("boxSize", C.c_float * 3),
                ("fourier", _U16P), ("wFirst", _U16P), ("bFirst", _U16P), ("wHidden", _U16P), ("bHidden", _U16P),
                ("wLast", _U16P), ("bLast", _U16P), ("gridX", C.c_int), ("gridY", C.c_int), ("gridZ", C.c_int),
                ("gridTexA", C.POINTER(C.c_void_p)), ("gridTexB", C.POINTER(C.c_void_p)),
                ("gridOffsetA", _FP), ("gridScaleA", _FP), ("gridInterpolation", _FP)]


class _OracleScene(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3), ("fovY", C.c_float),
                ("stepsize", C.c_float), ("densityMin", C.c_float), ("densityMax", C.c_float),
                ("earlyOut", C.c_int), ("blendMode", C.c_int), ("tfKind", C.c_int), ("tfRows", C.c_int),
                ("tfScaleAbsorption", C.c_float), ("tfScaleEmission", C.c_float), ("tfTable", _FP),
                ("gradientMode", C.c_int), ("fdStep", C.c_float), ("gridDiffStep", C.c_float),
                ("brdfMagnitudeScaling", C.c_int), ("brdfPhong", C.c_int), ("brdfLightType", C.c_int),
                ("brdfSpecularExponent", C.c_int), ("brdfMagScale", C.c_float), ("brdfAmbient", C.c_float),
                ("brdfSpecular", C.c_float), ("brdfMagCenter", C.c_float), ("brdfMagRadius", C.c_float),
                ("brdfLight", C.c_float * 3), ("tfPreintegration", C.c_int), ("tfPreintegrated", _FP), ("tfGaussianMode", C.c_int),
                ("rotationResync", C.c_int), ("segments", C.c_int), ("rotationHiLo", C.c_int)]


_lib = None


def build() -> None:
    """Compiles the C restatement with gcc (test infrastructure; `make -C oracle`)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


VOLUME_NEAREST, VOLUME_TRILINEAR, VOLUME_TRICUBIC = range(3)
VOLUME_SOURCE_TEXTURE, VOLUME_SOURCE_TENSOR = range(2)


class _OracleVolume(C.Structure):
    _fields_ = [("data", _FP), ("res", C.c_int * 3), ("boxMin", C.c_float * 3), ("boxSize", C.c_float * 3),
                ("interpolation", C.c_int), ("source", C.c_int), ("newBehavior", C.c_int), ("provideNormals", C.c_int)]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.oracle_eval_points.restype = C.c_int
        l.oracle_eval_points.argtypes = [C.POINTER(_OracleNet), _FP, _FP, C.c_size_t, _FP]
        l.oracle_eval_adjoint.restype = C.c_int
        l.oracle_eval_adjoint.argtypes = [C.POINTER(_OracleNet), _FP, _FP, C.c_size_t, C.c_float, _FP]
        l.oracle_tf_evaluate.restype = None
        l.oracle_tf_evaluate.argtypes = [C.POINTER(_OracleScene), _FP, _FP, C.c_size_t, _FP]
        l.oracle_tf_preintegrate.restype = None
        l.oracle_tf_preintegrate.argtypes = [_FP, C.c_int, C.c_int, C.c_float, C.c_int, _FP]
        l.oracle_eval_points_full.restype = C.c_int
        l.oracle_eval_points_full.argtypes = [C.POINTER(_OracleNet), _FP, _FP, C.c_size_t, _FP]
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.POINTER(_OracleNet), C.POINTER(_OracleScene), C.c_int, C.c_int, C.c_int, C.c_int,
                                    _FP, C.POINTER(C.c_ulonglong)]
        l.oracle_count_samples.restype = C.c_ulonglong
        l.oracle_count_samples.argtypes = [C.POINTER(_OracleNet), C.POINTER(_OracleScene), C.c_int, C.c_int, C.c_int, C.c_int]
        l.oracle_volume_eval_points.restype = None
        l.oracle_volume_eval_points.argtypes = [C.POINTER(_OracleVolume), _FP, C.c_size_t, _FP]
        l.oracle_render_volume.restype = C.c_int
        l.oracle_render_volume.argtypes = [C.POINTER(_OracleVolume), C.POINTER(_OracleScene), C.c_int, C.c_int, _FP, C.POINTER(C.c_ulonglong)]
        l.oracle_float_to_half.restype = C.c_uint16
        l.oracle_float_to_half.argtypes = [C.c_float]
        l.oracle_half_to_float.restype = C.c_float
        l.oracle_half_to_float.argtypes = [C.c_uint16]
        _lib = l
    return _lib


def _u16(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint16)


class OracleNetwork:
    """The reference constant block of one SceneNetwork (a volnet_io.VolnetData) at (time, ensemble)."""

    def __init__(self, vn, acc_mode: int = ACC_HALF, time: float = 0.0, ensemble: int = 0):
        n = _OracleNet()
        keep = []
        F = vn.num_fourier
        has_grid = vn.has_grid()
        if F <= 0 and has_grid:
            raise ValueError("a latent grid needs Fourier features (SceneNetwork::valid, volume_interpolation_network.cpp:952-956)")
        L = vn.layers
        G = 0
        if has_grid:
            tg, eg = vn.time_grids or [], vn.ensemble_grids or []
            G = (tg[0].channels if tg else 0) + (eg[0].channels if eg else 0)
        # getDefines (:1139-1219)
        base = 8 if vn.has_direction else 4
        if F > 0:
            Cc = L[0].channels_in - G
            assert Cc == base + 2 * F, "2*num_fourier+%d == hidden[0].channelsIn must hold" % base
        else:  # scalar first layer 3|6 -> C (renderer_volume_tensorcores.cuh:810-823)
            Cc = L[0].channels_out
            assert L[0].channels_in == (6 if vn.has_direction else 3)
        n.useDirection = (2 if vn.use_direction_in_fourier else 1) if vn.has_direction else 0
        start_hidden = 1 if (has_grid or F <= 0) else 0
        NH = len(L) - start_hidden - 1
        n.C, n.F, n.G, n.NH = Cc, F, G, NH
        n.Cout = L[-1].channels_out
        n.outputMode = OUTPUT_MODES[vn.output_mode]
        n.activation = ACTIVATIONS[L[0].activation]
        n.actParam = L[0].activation_param
        n.passTime = int(vn.has_time)
        n.accMode = acc_mode
        n.boxMin[:] = list(vn.box_min)
        n.boxSize[:] = list(vn.box_size)

        def ptr16(a):
            a = _u16(a)
            keep.append(a)
            return a.ctypes.data_as(_U16P)

        n.fourier = ptr16(vn.fourier if F > 0 else np.zeros(1, np.uint16))
        if has_grid or F <= 0:
            n.wFirst = ptr16(L[0].weights)
            n.bFirst = ptr16(L[0].bias)
        hidden = L[start_hidden:len(L) - 1]
        n.wHidden = ptr16(np.concatenate([_u16(l.weights) for l in hidden]) if hidden else np.zeros(1, np.uint16))
        n.bHidden = ptr16(np.concatenate([_u16(l.bias) for l in hidden]) if hidden else np.zeros(1, np.uint16))
        n.wLast = ptr16(L[-1].weights)  # stored [cin][cout] (addLayer transposes small layers)
        n.bLast = ptr16(L[-1].bias)

        if has_grid:  # :1289-1362
            tg, eg = vn.time_grids or [], vn.ensemble_grids or []
            first = (tg or eg)[0]
            n.gridEncoding = first.encoding
            n.gridX, n.gridY, n.gridZ = first.size_x, first.size_y, first.size_z
            texA, texB, off, sc, interp = [], [], [], [], []

            def tex_ptr(g, i):
                a = np.ascontiguousarray(g.data[i])
                keep.append(a)
                return a.ctypes.data

            if tg:
                tnum = len(tg)
                # setTimeAndEnsemble clamps to the key-frame range (:923-938), interpolateTime (.h:353-357)
                tmax_incl = vn.time_min + (tnum - 1) * vn.time_step
                t = min(max(float(time), float(vn.time_min)), float(tmax_incl))
                tt = np.float32((np.float32(t) - np.float32(vn.time_min)) / np.float32(vn.time_step))
                tt = float(min(max(tt, np.float32(0)), np.float32(tnum - 1)))
                lo = min(int(tt), tnum - 1)
                hi = min(lo + 1, tnum - 1)
                for i in range(tg[0].channels // 4):
                    texA.append(tex_ptr(tg[lo], i))
                    texB.append(tex_ptr(tg[hi], i))
                    interp.append(tt)
                if first.encoding != 0:
                    off += list(tg[lo].offset)
                    sc += list(tg[lo].scale)
                else:
                    off += [0.0] * tg[0].channels
                    sc += [1.0] * tg[0].channels
            if eg:
                idx = min(max(int(ensemble) - vn.ensemble_min, 0), len(eg) - 1)
                for i in range(eg[0].channels // 4):
                    texA.append(tex_ptr(eg[idx], i))
                    texB.append(tex_ptr(eg[idx], i))
                    interp.append(0.0)
                if first.encoding != 0:
                    off += list(eg[idx].offset)
                    sc += list(eg[idx].scale)
                else:
                    off += [0.0] * eg[0].channels
                    sc += [1.0] * eg[0].channels
            ta = (C.c_void_p * len(texA))(*texA)
            tb = (C.c_void_p * len(texB))(*texB)
            oa = np.asarray(off, np.float32)
            sa = np.asarray(sc, np.float32)
            ia = np.asarray(interp, np.float32)
            keep += [ta, tb, oa, sa, ia]
            n.gridTexA = C.cast(ta, C.POINTER(C.c_void_p))
            n.gridTexB = C.cast(tb, C.POINTER(C.c_void_p))
            n.gridOffsetA = oa.ctypes.data_as(_FP)
            n.gridScaleA = sa.ctypes.data_as(_FP)
            n.gridInterpolation = ia.ctypes.data_as(_FP)
        self._n = n
        self._keep = keep
        self.output_channels = 4 if vn.output_mode in ("rgbo", "rgbo:direct") else 1

    def evaluate(self, world_positions: np.ndarray, directions: Optional[np.ndarray] = None) -> np.ndarray:
        p = np.ascontiguousarray(world_positions, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32) if directions is not None else None
        out = np.zeros((p.shape[0], self.output_channels), np.float32)
        r = lib().oracle_eval_points(C.byref(self._n), p.ctypes.data_as(_FP), d.ctypes.data_as(_FP) if d is not None else None,
                                     p.shape[0], out.ctypes.data_as(_FP))
        if r != 0:
            raise RuntimeError("oracle_eval_points failed")
        return out


    def adjoint_gradient(self, world_positions: np.ndarray, directions: Optional[np.ndarray] = None, grid_step: float = 0.0) -> np.ndarray:
        """_evalNormalAdjoint (renderer_volume_tensorcores.cuh:1198-1540): (n,3) gradients of the un-clamped density w.r.t. the
        NORMALIZED position; grid_step = 0: the reference's default 1 / (grid resolution * 4)."""
        p = np.ascontiguousarray(world_positions, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32) if directions is not None else None
        if grid_step <= 0:
            grid_step = 1.0 / (max(int(self._n.gridX), 1) * 4.0)
        out = np.zeros((p.shape[0], 3), np.float32)
        r = lib().oracle_eval_adjoint(C.byref(self._n), p.ctypes.data_as(_FP), d.ctypes.data_as(_FP) if d is not None else None,
                                      p.shape[0], grid_step, out.ctypes.data_as(_FP))
        if r != 0:
            raise RuntimeError("oracle_eval_adjoint failed")
        return out

    def evaluate_full(self, world_positions: np.ndarray, directions: Optional[np.ndarray] = None) -> np.ndarray:
        """(N,9): value[4], predicted normal[3], predicted curvature[2] -- everything eval<> returns."""
        p = np.ascontiguousarray(world_positions, dtype=np.float32)
        d = np.ascontiguousarray(directions, dtype=np.float32) if directions is not None else None
        out = np.zeros((p.shape[0], 9), np.float32)
        r = lib().oracle_eval_points_full(C.byref(self._n), p.ctypes.data_as(_FP), d.ctypes.data_as(_FP) if d is not None else None,
                                          p.shape[0], out.ctypes.data_as(_FP))
        if r != 0:
            raise RuntimeError("oracle_eval_points_full failed")
        return out


class OracleVolume:
    """kernel::VolumeInterpolationGrid on a dense (X,Y,Z) array (renderer/renderer_volume_grid.cuh); u8 / u16 arrays are read as
    normalised floats like the reference's textures."""

    def __init__(self, data_xyz: np.ndarray, box_min, box_size, interpolation=VOLUME_TRILINEAR, source=VOLUME_SOURCE_TEXTURE,
                 new_behavior=False, provide_normals=False):
        a = np.asarray(data_xyz)
        if a.dtype == np.uint8:
            a = a.astype(np.float32) / np.float32(255.0)
        elif a.dtype == np.uint16:
            a = a.astype(np.float32) / np.float32(65535.0)
        self._d = np.ascontiguousarray(a.astype(np.float32).transpose(2, 1, 0))  # [z][y][x]: x fastest
        v = _OracleVolume()
        v.data = self._d.ctypes.data_as(_FP)
        v.res[:] = [int(n) for n in a.shape]
        v.boxMin[:] = [float(x) for x in box_min]
        v.boxSize[:] = [float(x) for x in box_size]
        v.interpolation, v.source, v.newBehavior = int(interpolation), int(source), int(new_behavior)
        v.provideNormals = int(provide_normals)
        self._v = v

    def evaluate(self, positions: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        out = np.zeros(p.shape[0], np.float32)
        lib().oracle_volume_eval_points(C.byref(self._v), p.ctypes.data_as(_FP), p.shape[0], out.ctypes.data_as(_FP))
        return out

    def render(self, scene: "OracleScene", width: int, height: int):
        out = np.zeros((8, height, width), np.float32)
        cnt = C.c_ulonglong(0)
        if lib().oracle_render_volume(C.byref(self._v), C.byref(scene._s), width, height, out.ctypes.data_as(_FP), C.byref(cnt)) != 0:
            raise RuntimeError("oracle_render_volume failed")
        return out, cnt.value


class OracleScene:
    def __init__(self, *, eye, right, up, fov_y_radians, stepsize, density_min=0.0, density_max=1.0, early_out=True,
                 blend_mode=BLEND_BEER_LAMBERT, tf_kind=TF_IDENTITY, tf_scale_absorption=1.0, tf_scale_emission=1.0,
                 tf_table=None, gradient_mode=0, finite_differences_stepsize=0.0, brdf=None, tf_preintegration=0,
                 adjoint_grid_stepsize=0.0, tf_gaussian_mode=0, rotation_resync=0, segments=1, rotation_hilo=1):
        """rotation_resync / segments / rotation_hilo: only read by the ACC_DEVICE model (see srn_oracle.h, OracleScene); rotation_hilo = 1 is
        what the kernels do since r04 (re-derivations from the fp32 position), 0 the r01 - r03 statement (from its fp16 rounding)."""
        s = _OracleScene()
        s.tfGaussianMode = tf_gaussian_mode
        assert rotation_resync >= 0 and (rotation_resync & (rotation_resync - 1)) == 0, "resync period: 0 or a power of two"
        s.rotationResync, s.segments = int(rotation_resync), int(segments)
        s.rotationHiLo = int(rotation_hilo)
        s.eye[:] = [float(v) for v in eye]
        s.right[:] = [float(v) for v in right]
        s.up[:] = [float(v) for v in up]
        s.fovY = fov_y_radians
        s.stepsize = stepsize
        s.densityMin, s.densityMax = density_min, density_max
        s.earlyOut, s.blendMode, s.tfKind = int(early_out), blend_mode, tf_kind
        s.tfScaleAbsorption, s.tfScaleEmission = tf_scale_absorption, tf_scale_emission
        self._t = None
        if tf_table is not None:
            self._t = np.ascontiguousarray(tf_table, dtype=np.float32)
            s.tfTable = self._t.ctypes.data_as(_FP)
            s.tfRows = self._t.shape[0]
        self._p = None
        if tf_preintegration:
            R = self._t.shape[0]
            self._p = np.zeros((R, 4) if tf_preintegration == 1 else (R, R, 4), np.float32)
            lib().oracle_tf_preintegrate(self._t.ctypes.data_as(_FP), R, tf_preintegration, stepsize, 256, self._p.ctypes.data_as(_FP))
            s.tfPreintegration = tf_preintegration
            s.tfPreintegrated = self._p.ctypes.data_as(_FP)
        s.gradientMode, s.fdStep = gradient_mode, finite_differences_stepsize
        self._adjoint_grid_stepsize = adjoint_grid_stepsize
        if brdf:  # same keys as capi.Scene
            s.brdfPhong = int(brdf.get("enable_phong", False))
            s.brdfMagnitudeScaling = int(brdf.get("enable_magnitude_scaling", False))
            s.brdfMagScale = brdf.get("magnitude_scaling", 1.0)
            s.brdfAmbient = brdf.get("ambient", 0.1)
            s.brdfSpecular = brdf.get("specular", 0.1)
            s.brdfMagCenter = brdf.get("magnitude_center", 0.5)
            s.brdfMagRadius = brdf.get("magnitude_radius", 0.1)
            s.brdfSpecularExponent = int(brdf.get("specular_exponent", 16))
            s.brdfLightType = int(brdf.get("light_type", 0))
            s.brdfLight[:] = [float(v) for v in brdf.get("light", (0.0, 0.0, 1.0))]
        self._s = s

    def render(self, net: OracleNetwork, width: int, height: int, y0: int = 0, y1: Optional[int] = None):
        """-> ((8,H,W) fp32 image, evaluated sample count)"""
        if y1 is None:
            y1 = height
        out = np.zeros((8, height, width), np.float32)
        cnt = C.c_ulonglong(0)
        # latentGridDifferencesStepSize of the adjoint mode: 1 / (grid resolution * 4) unless given (volume_interpolation_network.cpp:1808-1812)
        self._s.gridDiffStep = self._adjoint_grid_stepsize if self._adjoint_grid_stepsize > 0 else 1.0 / (max(int(net._n.gridX), 1) * 4.0)
        r = lib().oracle_render(C.byref(net._n), C.byref(self._s), width, height, y0, y1, out.ctypes.data_as(_FP), C.byref(cnt))
        if r != 0:
            raise RuntimeError("oracle_render failed")
        return out, cnt.value

    def evaluate_tf(self, densities: np.ndarray, previous: Optional[np.ndarray] = None) -> np.ndarray:
        """EvaluateTF / EvaluateTFWithPrevious with this scene's TF, density range and step size: (n,) -> (n,4)."""
        d = np.ascontiguousarray(densities, np.float32).reshape(-1)
        p = np.ascontiguousarray(previous, np.float32).reshape(-1) if previous is not None else None
        out = np.zeros((d.size, 4), np.float32)
        lib().oracle_tf_evaluate(C.byref(self._s), d.ctypes.data_as(_FP), p.ctypes.data_as(_FP) if p is not None else None, d.size,
                                 out.ctypes.data_as(_FP))
        return out

    def count_samples(self, net: OracleNetwork, width: int, height: int, y0: int = 0, y1: Optional[int] = None) -> int:
        if y1 is None:
            y1 = height
        return int(lib().oracle_count_samples(C.byref(net._n), C.byref(self._s), width, height, y0, y1))


def camera_on_a_sphere(orientation: str, center, pitch: float, yaw: float, distance: float):
    """numpy float64 restatement of CameraOnASphere (renderer/camera.cpp:17-35,458-490,553-581)."""
    names = ["Xp", "Xm", "Yp", "Ym", "Zp", "Zm"]
    ups = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    perms = [(2, -1, -3), (-2, 1, 3), (1, 2, 3), (-1, -2, -3), (-3, -1, 2), (3, 1, -2)]
    inv_yaw = [False, True, True, False, True, False]
    o = names.index(orientation)
    yaw = -yaw if not inv_yaw[o] else yaw
    pitch = -pitch
    pos = np.array([np.cos(pitch) * np.cos(yaw) * distance, np.sin(pitch) * distance, np.cos(pitch) * np.sin(yaw) * distance])
    pos2 = np.array([pos[abs(p) - 1] * (1 if p > 0 else -1) for p in perms[o]])
    look_at = np.asarray(center, np.float64)
    origin = pos2 + look_at
    front = look_at - origin
    front /= np.linalg.norm(front)
    right = np.cross(front, np.asarray(ups[o], np.float64))
    right /= np.linalg.norm(right)
    up = np.cross(right, front)
    up /= np.linalg.norm(up)
    return origin.astype(np.float32), right.astype(np.float32), up.astype(np.float32)


# ---- IImageEvaluator::ExtractColor (renderer/iimage_evaluator.cpp:26-135, iimage_evaluator_cuda.cu:82-101,144-165) ----
CHANNEL_MASK, CHANNEL_NORMAL, CHANNEL_DEPTH, CHANNEL_COLOR = range(4)


def depth_range(raw: np.ndarray) -> np.ndarray:
    """{-min, max, nan flag} of the depth plane of an image part (the mergeable form of fvsrn_depth_range): min / max over the finite-or-infinite values,
    the flag says whether any NaN was seen; merged over parts by an element-wise maximum."""
    d = np.asarray(raw, np.float32)[7].reshape(-1)
    nan = np.isnan(d)
    v = d[~nan]
    if v.size == 0:
        return np.array([-np.inf, -np.inf, 1.0 if nan.any() else 0.0], np.float32)
    return np.array([-v.min(), v.max(), 1.0 if nan.any() else 0.0], np.float32)


def extract_color(raw: np.ndarray, channel_mode: int = CHANNEL_COLOR, use_tonemapping: bool = False,
                  max_exposure: float = 1.0, depth_range3=None) -> np.ndarray:
    """raw (8,H,W) fp32 -> (4,H,W) fp32, numpy restatement (test infrastructure).  depth_range3: the merged {-min, max, nan flag} of the whole frame when
    `raw` is one rank's part of it (CHANNEL_DEPTH)."""
    raw = np.asarray(raw, np.float32)
    f = np.float32
    if channel_mode == CHANNEL_COLOR:
        if not use_tonemapping:
            return raw[:4].copy()
        rgb = raw[:3] / f(max_exposure)
        with np.errstate(invalid="ignore", divide="ignore"):
            rgb = (rgb * (f(2.51) * rgb + f(0.03))) / (rgb * (f(2.43) * rgb + f(0.59)) + f(0.14))
            rgb = np.power(np.clip(rgb, 0, 1), f(1.0 / 2.4)).astype(np.float32)
        return np.concatenate([rgb, raw[3:4]], axis=0)
    if channel_mode == CHANNEL_DEPTH:
        d = raw[7]
        with np.errstate(invalid="ignore", divide="ignore"):
            mn, mx = d.min(), d.max()  # numpy, like torch, propagates NaN
            if depth_range3 is not None:
                r3 = np.asarray(depth_range3, np.float32)
                mn, mx = (np.float32(np.nan), np.float32(np.nan)) if r3[2] != 0 else (-r3[0], r3[1])
            scale, offset = f(1) / (mx - mn), -mn / (mx - mn)
            v = d * scale + offset
        return np.stack([v, v, v, np.ones_like(v)], axis=0).astype(np.float32)
    if channel_mode == CHANNEL_MASK:
        a = raw[3]
        return np.stack([a, a, a, np.ones_like(a)], axis=0)
    if channel_mode == CHANNEL_NORMAL:
        return np.concatenate([raw[4:7] * f(0.5) + f(0.5), raw[3:4]], axis=0).astype(np.float32)
    raise ValueError("unknown channel mode")


def rgba_to_int(rgba: np.ndarray) -> np.ndarray:
    """rgbaToInt (renderer_utils.cuh:48-57) on a (4,H,W) image: uint32 words 0xAABBGGRR."""
    with np.errstate(invalid="ignore"):
        q = np.clip(np.asarray(rgba, np.float32) * np.float32(255), 0, 255)
        q = np.where(np.isnan(q), 0, q).astype(np.uint32)  # fminf(fmaxf(NaN, 0), 255) = 0 on the device
    return (q[3] << 24) | (q[2] << 16) | (q[1] << 8) | q[0]


# ---- transfer functions from their scene-file description (host-side table construction of the reference) ----
def tf_piecewise_table(color_points, opacity_points, absorption_scaling: float) -> np.ndarray:
    """TransferFunctionPiecewiseLinear::computeTensor (renderer/transfer_function_piecewise.cpp:166-282):
    colorPoints [[pos,r,g,b]...], opacityPoints [[pos,absorption]...] -> (R,5) rows [r,g,b,absorption*scaling,pos]."""
    col = sorted(([float(p[0]), np.asarray(p[1:4], np.float64)] for p in color_points), key=lambda p: p[0])
    opa = sorted(([float(p[0]), float(p[1])] for p in opacity_points), key=lambda p: p[0])
    if col[0][0] > 0:
        col.insert(0, [-1.0, col[0][1]])
    if opa[0][0] > 0:
        opa.insert(0, [-1.0, opa[0][1]])
    if col[-1][0] < 1:
        col.append([2.0, col[-1][1]])
    if opa[-1][0] < 1:
        opa.append([2.0, opa[-1][1]])
    pts = [[min(col[0][0], opa[0][0]) if col[0][0] <= opa[0][0] else opa[0][0], col[0][1], opa[0][1]]]
    io = ic = 0
    while io < len(opa) - 1 and ic < len(col) - 1:
        if opa[io + 1][0] < col[ic + 1][0]:
            f = (opa[io + 1][0] - col[ic][0]) / (col[ic + 1][0] - col[ic][0])
            pts.append([opa[io + 1][0], col[ic][1] + f * (col[ic + 1][1] - col[ic][1]), opa[io + 1][1]])
            io += 1
        else:
            f = (col[ic + 1][0] - opa[io][0]) / (opa[io + 1][0] - opa[io][0])
            pts.append([col[ic + 1][0], col[ic + 1][1], opa[io][1] + f * (opa[io + 1][1] - opa[io][1])])
            ic += 1
    eps = float(np.float32(1e-7))
    i = 0
    while i < len(pts) - 2:
        if (pts[i][2] < eps and pts[i + 1][2] < eps and pts[i + 2][2] < eps) or (pts[i + 1][0] - pts[i][0] < eps):
            del pts[i + 1]
        else:
            i += 1
    out = np.zeros((len(pts), 5), np.float32)
    for k, (pos, rgb, a) in enumerate(pts):
        out[k, :3] = np.clip(rgb, 0.0, float(np.float32(1.0) - np.finfo(np.float32).eps))
        out[k, 3] = min(max(a, 0.0), 1.0) * absorption_scaling
        out[k, 4] = pos
    return out


def tf_texture_table(color_points, opacity_plot, absorption_scaling: float) -> np.ndarray:
    """TransferFunctionTexture::computeTexture (renderer/transfer_function_texture.cpp:347-362) +
    TFPartPiecewiseColor::getAsTexture (transfer_function.cpp:526-551): (256,4) texels [r,g,b,scaling*plot[i]]."""
    col = sorted(([float(p[0]), np.asarray(p[1:4], np.float64)] for p in color_points), key=lambda p: p[0])
    R = 256
    assert len(opacity_plot) == R
    out = np.zeros((R, 4), np.float32)
    n = len(col)
    for i in range(R):
        density = np.float32((i + np.float32(0.5)) / np.float32(R))
        idx = 0
        while idx < n - 2 and not (col[idx + 1][0] > density):
            idx += 1
        lo, hi = col[idx], col[min(idx + 1, n - 1)]
        with np.errstate(invalid="ignore", divide="ignore"):
            frac = np.clip(np.float32((density - np.float32(lo[0])) / (np.float32(hi[0]) - np.float32(lo[0]))), 0, 1)
        out[i, :3] = (1 - frac) * lo[1] + frac * hi[1]
        out[i, 3] = np.float32(absorption_scaling) * np.float32(opacity_plot[i])
    return out
