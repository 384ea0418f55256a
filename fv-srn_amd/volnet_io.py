"""
Pure-Python reader/writer of the ``.volnet`` container (no compute, no GPU).

Counterpart of the export half of the reference's ``SceneRepresentationNetwork.export_to_pyrenderer``
+ ``SceneNetwork.save`` (applications/volnet/network.py:798-897,
renderer/volume_interpolation_network.cpp:1088-1104): it lets a training script write a network
without going through the C++ module, and gives the tests a second, independent implementation of
the on-disk format (little endian, layout documented in SURVEY.md 8(b)) to cross-check the C++
reader/writer in csrc/scene_network.cpp.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import List, Optional, Sequence

import numpy as np

OUTPUT_MODES = ["density", "density:direct", "rgbo", "rgbo:direct", "densitygrad", "densitygrad:direct",
                "densitygrad:cubic", "densitycurvature", "densitycurvature:direct"]
OUTPUT_MODE_CHANNELS_IN = [1, 1, 4, 4, 4, 4, 4, 6, 6]
ACTIVATIONS = ["ReLU", "Sine", "Snake", "SnakeAlt", "Sigmoid", "None"]
ENC_FLOAT, ENC_BYTE_LINEAR, ENC_BYTE_GAUSSIAN = 0, 1, 2


def to_half_bits(a) -> np.ndarray:
    """fp32 -> IEEE half bits (round to nearest even), like __float2half on the host."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


@dataclasses.dataclass
class LayerData:
    """One Linear layer AS STORED by SceneNetwork::addLayer (padded / transposed, half bits)."""
    channels_out: int
    channels_in: int
    weights: np.ndarray  # uint16, channels_out*channels_in entries
    bias: np.ndarray     # uint16, channels_out
    activation: str
    activation_param: float = 1.0


@dataclasses.dataclass
class GridData:
    encoding: int
    channels: int
    size_z: int
    size_y: int
    size_x: int
    data: np.ndarray  # float32 or uint8, layout [C/4][Z][Y][X][4]
    offset: Optional[np.ndarray] = None  # float32 [C] (byte encodings)
    scale: Optional[np.ndarray] = None


@dataclasses.dataclass
class VolnetData:
    has_time: bool = False
    has_direction: bool = False
    num_fourier: int = 0
    use_direction_in_fourier: bool = False
    fourier: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, np.uint16))  # feature fastest
    output_mode: str = "density"
    layers: List[LayerData] = dataclasses.field(default_factory=list)
    box_min: Sequence[float] = (-5.0, -5.0, -5.0)
    box_size: Sequence[float] = (1.0, 1.0, 1.0)
    time_min: int = 0
    time_step: int = 1
    ensemble_min: int = 0
    time_grids: Optional[List[GridData]] = None
    ensemble_grids: Optional[List[GridData]] = None

    def has_grid(self) -> bool:
        return self.time_grids is not None or self.ensemble_grids is not None


def _pack_str(s: str) -> bytes:
    b = s.encode("ascii")
    return struct.pack("<i", len(b)) + b


def fourier_to_stored(B: np.ndarray, premultiplied: bool = True) -> np.ndarray:
    """(F, 3|6) fp32 -> stored half bits, feature fastest (volume_interpolation_network.cpp:129-156)."""
    B = np.asarray(B, dtype=np.float32)
    v = (1.0 if premultiplied else 2 * np.pi) * B.astype(np.float64)
    return to_half_bits(v.astype(np.float32).T.reshape(-1))


def add_layer(net: VolnetData, weight: np.ndarray, bias: np.ndarray, activation: str, param: float = 1.0) -> None:
    """SceneNetwork::addLayerFromTorch + addLayer (volume_interpolation_network.cpp:806-921)."""
    w = np.asarray(weight, dtype=np.float32)
    cout, cin = w.shape
    wh = to_half_bits(w).reshape(cout, cin)
    bh = to_half_bits(bias).reshape(cout)
    if not net.layers and net.num_fourier > 0:
        zero = np.zeros((cout, 1), np.uint16)
        if not net.has_time:
            if net.has_direction:
                wh = np.concatenate([wh[:, :3], zero, wh[:, 3:6], zero, wh[:, 6:]], axis=1)
            else:
                wh = np.concatenate([wh[:, :3], zero, wh[:, 3:]], axis=1)
        elif net.has_direction:
            wh = np.concatenate([wh[:, :7], zero, wh[:, 7:]], axis=1)
        stored = wh
    elif cin < 16 or cout < 16:
        stored = wh.T  # [in][out]
    else:
        stored = wh
    net.layers.append(LayerData(cout, stored.size // cout, np.ascontiguousarray(stored).reshape(-1), bh,
                                activation, float(param)))


def encode_grid(grid: np.ndarray, encoding: int) -> GridData:
    """LatentGrid(tensor, encoding) (volume_interpolation_network.cpp:290-431); grid is (C,Z,Y,X) fp32."""
    g = np.asarray(grid, dtype=np.float32)
    C, Z, Y, X = g.shape
    assert C % 4 == 0
    def relayout(a):
        return np.ascontiguousarray(a.reshape(C // 4, 4, Z, Y, X).transpose(0, 2, 3, 4, 1))
    if encoding == ENC_FLOAT:
        return GridData(encoding, C, Z, Y, X, relayout(g))
    if encoding == ENC_BYTE_LINEAR:
        mn = g.reshape(C, -1).min(axis=1)
        mx = g.reshape(C, -1).max(axis=1)
        inv = (np.float32(1.0) / np.maximum(np.float32(1e-5), mx - mn)).astype(np.float32)
        x01 = (g - mn[:, None, None, None]) * inv[:, None, None, None]
        q = np.clip(_roundf(np.float32(255) * x01), 0, 255).astype(np.uint8)
        return GridData(encoding, C, Z, Y, X, relayout(q), mn.astype(np.float32), (mx - mn).astype(np.float32))
    if encoding == ENC_BYTE_GAUSSIAN:
        from math import erf
        flat = g.reshape(C, -1).astype(np.float64)
        mean = flat.mean(axis=1)
        std = flat.std(axis=1, ddof=1)
        meanf, stdf = mean.astype(np.float32), std.astype(np.float32)
        inv = (np.float32(1.0) / np.maximum(np.float32(1e-5), stdf)).astype(np.float32)
        hat = (g - meanf[:, None, None, None]) * inv[:, None, None, None]
        verf = np.vectorize(erf)
        theta = (np.float32(0.5) * (np.float32(1) + verf((hat * np.float32(0.70710678118654752440)).astype(np.float32))
                                    .astype(np.float32))).astype(np.float32)
        q = np.clip(_roundf(np.float32(255) * theta), 0, 255).astype(np.uint8)
        return GridData(encoding, C, Z, Y, X, relayout(q), meanf, stdf)
    raise ValueError("Unsupported encoding")


def _roundf(a: np.ndarray) -> np.ndarray:
    """C roundf: halves away from zero (np.round rounds halves to even)."""
    a = np.asarray(a, dtype=np.float32)
    return np.where(a >= 0, np.floor(a + np.float32(0.5)), np.ceil(a - np.float32(0.5)))


def save_volnet(net: VolnetData) -> bytes:
    out = [struct.pack("<i", 2)]  # SceneNetwork::VERSION
    # InputParametrization v3 (:116-127)
    out.append(struct.pack("<i??i?", 3, net.has_time, net.has_direction, net.num_fourier, net.use_direction_in_fourier))
    out.append(np.asarray(net.fourier, np.uint16).tobytes())
    # OutputParametrization v1 (:217-221)
    out.append(struct.pack("<i", 1) + _pack_str(net.output_mode))
    out.append(struct.pack("<i", len(net.layers)))
    for l in net.layers:  # Layer v2 (:274-288)
        out.append(struct.pack("<iii", 2, l.channels_out, l.channels_in))
        out.append(np.asarray(l.weights, np.uint16).tobytes())
        out.append(np.asarray(l.bias, np.uint16).tobytes())
        out.append(_pack_str(l.activation) + struct.pack("<f", l.activation_param))
    out.append(struct.pack("<6f", *net.box_min, *net.box_size))
    out.append(struct.pack("<b", 1 if net.has_grid() else 0))
    if net.has_grid():  # LatentGridTimeAndEnsemble v1 (:782-796)
        tg = net.time_grids or []
        eg = net.ensemble_grids or []
        out.append(struct.pack("<6i", 1, net.time_min, len(tg), net.time_step, net.ensemble_min, len(eg)))
        for g in list(tg) + list(eg):  # LatentGrid v1 (:595-614)
            out.append(struct.pack("<6i", 1, g.encoding, g.channels, g.size_z, g.size_y, g.size_x))
            out.append(np.ascontiguousarray(g.data).tobytes())
            if g.encoding != ENC_FLOAT:
                out.append(np.asarray(g.offset, np.float32).tobytes())
                out.append(np.asarray(g.scale, np.float32).tobytes())
    return b"".join(out)


class _Reader:
    def __init__(self, b: bytes):
        self.b, self.p = b, 0

    def take(self, fmt: str):
        size = struct.calcsize(fmt)
        if self.p + size > len(self.b):
            raise ValueError("unexpected end of .volnet data")
        v = struct.unpack_from(fmt, self.b, self.p)
        self.p += size
        return v

    def array(self, dtype, count: int) -> np.ndarray:
        nbytes = np.dtype(dtype).itemsize * count
        if self.p + nbytes > len(self.b):
            raise ValueError("unexpected end of .volnet data")
        a = np.frombuffer(self.b, dtype=dtype, count=count, offset=self.p).copy()
        self.p += nbytes
        return a

    def string(self) -> str:
        (n,) = self.take("<i")
        s = self.b[self.p:self.p + n].decode("ascii")
        self.p += n
        return s


def load_volnet(data: bytes) -> VolnetData:
    r = _Reader(data)
    (version,) = r.take("<i")
    if version not in (1, 2):
        raise ValueError("Unknown version for SceneNetwork %d" % version)
    net = VolnetData()
    (iv,) = r.take("<i")
    if iv == 1:
        net.has_direction, net.num_fourier = r.take("<?i")
    elif iv == 2:
        net.has_direction, net.num_fourier, net.use_direction_in_fourier = r.take("<?i?")
    elif iv == 3:
        net.has_time, net.has_direction, net.num_fourier, net.use_direction_in_fourier = r.take("<??i?")
    else:
        raise ValueError("Unknown version for InputParametrization %d" % iv)
    net.fourier = r.array(np.uint16, net.num_fourier * (6 if net.use_direction_in_fourier else 3))
    (ov,) = r.take("<i")
    if ov != 1:
        raise ValueError("Unknown version for OutputParametrization %d" % ov)
    net.output_mode = r.string()
    (nl,) = r.take("<i")
    for _ in range(nl):
        lv, rows, cols = r.take("<iii")
        w = r.array(np.uint16, rows * cols)
        b = r.array(np.uint16, rows)
        act = r.string()
        param = r.take("<f")[0] if lv == 2 else 1.0
        net.layers.append(LayerData(rows, cols, w, b, act, param))
    box = r.take("<6f")
    net.box_min, net.box_size = box[:3], box[3:]
    if version == 2:
        (has,) = r.take("<b")
        if has > 0:
            _, net.time_min, tn, net.time_step, net.ensemble_min, en = r.take("<6i")
            grids = []
            for _ in range(tn + en):
                _, enc, C, Z, Y, X = r.take("<6i")
                dt = np.float32 if enc == ENC_FLOAT else np.uint8
                d = r.array(dt, C * Z * Y * X).reshape(C // 4, Z, Y, X, 4)
                off = sc = None
                if enc != ENC_FLOAT:
                    off, sc = r.array(np.float32, C), r.array(np.float32, C)
                grids.append(GridData(enc, C, Z, Y, X, d, off, sc))
            net.time_grids, net.ensemble_grids = grids[:tn], grids[tn:]
    return net


def build_volnet(*, fourier_B: np.ndarray, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray],
                 activation: str, activation_param: float = 1.0, output_mode: str = "density",
                 box_min=(0.0, 0.0, 0.0), box_size=(1.0, 1.0, 1.0), premultiplied: bool = True,
                 time_grids: Optional[Sequence[np.ndarray]] = None, ensemble_grids: Optional[Sequence[np.ndarray]] = None,
                 grid_encoding: int = ENC_FLOAT, time_min: int = 0, time_step: int = 1, ensemble_min: int = 0,
                 has_time: bool = False, has_direction: bool = False) -> VolnetData:
    """The hand-off order of export_to_pyrenderer (network.py:877-890) on plain numpy arrays."""
    net = VolnetData(has_time=has_time, has_direction=has_direction, output_mode=output_mode, box_min=tuple(box_min), box_size=tuple(box_size),
                     time_min=time_min, time_step=time_step, ensemble_min=ensemble_min)
    B = np.asarray(fourier_B, np.float32)
    net.num_fourier = B.shape[0]
    net.use_direction_in_fourier = B.shape[1] == 6
    net.fourier = fourier_to_stored(B, premultiplied)
    if time_grids is not None or ensemble_grids is not None:
        net.time_grids = [encode_grid(g, grid_encoding) for g in (time_grids or [])]
        net.ensemble_grids = [encode_grid(g, grid_encoding) for g in (ensemble_grids or [])]
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        add_layer(net, w, b, activation if i < n - 1 else "None", activation_param if i < n - 1 else 1.0)
    return net
