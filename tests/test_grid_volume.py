"""Dense grid volumes (BASELINE.json configs[0]): VolumeInterpolationGrid behind the DVR loop.

CPU: the C restatement (oracle/srn_oracle.c, vol_*) against hand-computed properties of renderer_volume_grid.cuh:89-232 and
the .cvol container (volume.cpp:623-740).  GPU: fvsrn_volume_evaluate_points / fvsrn_render_volume against the restatement.
Parity of this row is restatement-pinned only: the reference evaluates grids in CUDA / its CPU kernels, none of which can be
built here."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import util  # noqa: E402
from oracle import oracle  # noqa: E402

BOX_MIN, BOX_SIZE = (-0.5, -0.4, -0.3), (1.0, 0.8, 0.6)


def make_volume(shape=(12, 10, 8), seed=3, dtype=np.float32):
    rng = np.random.RandomState(seed)
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    v = np.exp(-3 * (x * x + y * y + z * z)) + 0.15 * rng.rand(*shape)
    v = np.clip(v / v.max(), 0, 1)
    if dtype == np.uint8:
        return np.round(v * 255).astype(np.uint8)
    if dtype == np.uint16:
        return np.round(v * 65535).astype(np.uint16)
    return v.astype(np.float32)


def node_positions(shape, scale_minus_one=True):
    """world positions of the grid nodes under the old behaviour (object coordinate = index)"""
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    den = np.array([n - 1 if scale_minus_one else n for n in shape], np.float32)
    return np.array(BOX_MIN, np.float32) + idx / den * np.array(BOX_SIZE, np.float32), idx.astype(int)


def test_tensor_source_interpolates_the_nodes():
    data = make_volume()
    pos, idx = node_positions(data.shape)
    for interp in (oracle.VOLUME_NEAREST, oracle.VOLUME_TRILINEAR):
        v = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, interp, oracle.VOLUME_SOURCE_TENSOR)
        got = v.evaluate(pos)
        assert np.abs(got - data[idx[:, 0], idx[:, 1], idx[:, 2]]).max() < 2e-5  # positions are rounded to fp32


def test_texture_source_is_shifted_by_half_a_voxel():
    """tex3D filters around texel centres at +0.5 and the reference applies no offset: object coordinate
    i + 0.5 returns node i, the tensor branch returns the midpoint of nodes i and i + 1."""
    data = make_volume()
    X = data.shape[0]
    p = np.array([[BOX_MIN[0] + (3 + 0.5) / (X - 1) * BOX_SIZE[0], BOX_MIN[1], BOX_MIN[2]]], np.float32)
    tex = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRILINEAR, oracle.VOLUME_SOURCE_TEXTURE).evaluate(p)[0]
    ten = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRILINEAR, oracle.VOLUME_SOURCE_TENSOR).evaluate(p)[0]
    assert abs(tex - data[3, 0, 0]) < 5e-3   # 8-bit filter weights
    assert abs(ten - 0.5 * (data[3, 0, 0] + data[4, 0, 0])) < 1e-4


def test_cubic_reproduces_constants_and_is_smooth():
    const = np.full((6, 7, 8), 0.37, np.float32)
    rng = np.random.RandomState(1)
    pos = (np.array(BOX_MIN) + rng.rand(64, 3) * np.array(BOX_SIZE)).astype(np.float32)
    for src in (oracle.VOLUME_SOURCE_TEXTURE, oracle.VOLUME_SOURCE_TENSOR):
        got = oracle.OracleVolume(const, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRICUBIC, src).evaluate(pos)
        assert np.abs(got - 0.37).max() < 1e-5
    data = make_volume()
    v = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRICUBIC, oracle.VOLUME_SOURCE_TENSOR).evaluate(pos)
    assert data.min() - 0.05 <= v.min() and v.max() <= data.max() + 0.05  # B-spline smoothing stays inside the data range


def test_new_behaviour_scales_by_the_resolution():
    data = make_volume()
    pos, idx = node_positions(data.shape, scale_minus_one=False)
    got = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_NEAREST, oracle.VOLUME_SOURCE_TENSOR, new_behavior=True).evaluate(pos)
    assert np.abs(got - data[idx[:, 0], idx[:, 1], idx[:, 2]]).max() < 1e-6


def test_trilinear_restatement_matches_torch_grid_sample():
    """A third-party witness for the trilinear rows (the reference's CUDA / CPU grid kernels cannot be built here, so this row is restatement-pinned):
    torch.nn.functional.grid_sample on the same array.  Tensor source, old behaviour -- object coordinate = p (N - 1), nodes at integer coordinates
    (renderer_volume_grid.cuh:102-120) -- is grid_sample(align_corners=True); texture source, new behaviour -- p N with texel centres at + 0.5
    (:121-139) -- is grid_sample(align_corners=False) up to the 8 fractional bits of a CUDA texture unit's filter weights.
    Both with border padding = clamp addressing."""
    import torch
    data = make_volume((11, 9, 13), seed=8)
    rng = np.random.RandomState(4)
    unit = rng.rand(4096, 3).astype(np.float32)
    pos = (np.array(BOX_MIN, np.float32) + unit * np.array(BOX_SIZE, np.float32)).astype(np.float32)
    vol = torch.from_numpy(np.ascontiguousarray(data.transpose(2, 1, 0)))[None, None]  # (1, 1, D = z, H = y, W = x)
    grid = torch.from_numpy((pos - np.array(BOX_MIN, np.float32)) / np.array(BOX_SIZE, np.float32) * 2 - 1)[None, None, None]  # x -> W, y -> H, z -> D
    for source, new_behavior, align, tol in ((oracle.VOLUME_SOURCE_TENSOR, False, True, 2e-5), (oracle.VOLUME_SOURCE_TEXTURE, True, False, 4e-3)):
        want = torch.nn.functional.grid_sample(vol, grid, mode="bilinear", padding_mode="border", align_corners=align)[0, 0, 0, 0].numpy()
        got = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRILINEAR, source, new_behavior=new_behavior).evaluate(pos)
        assert np.abs(got - want).max() < tol, (source, new_behavior, float(np.abs(got - want).max()))
    assert want.std() > 0.05


def test_cvol_round_trip(tmp_path):
    """Volume::save / Volume::Volume(filename), uncompressed (volume.cpp:623-740): header layout and x-fastest voxel order."""
    from fvsrn_amd import capi
    for dtype in (np.uint8, np.uint16, np.float32):
        data = make_volume(dtype=dtype)
        path = str(tmp_path / ("v_%s.cvol" % np.dtype(dtype).name))
        capi.Volume.save_cvol(path, data, world_size=(1.0, 0.8, 0.6), feature_name="density")
        raw = open(path, "rb").read()
        assert raw[:4] == b"CVOL" and struct.unpack("<i", raw[4:8])[0] == 1
        assert struct.unpack("<3f", raw[8:20]) == pytest.approx((1.0, 0.8, 0.6))
        nfeat, flags = struct.unpack("<2i", raw[20:28])
        assert (nfeat, flags) == (1, 0)
        ln = struct.unpack("<i", raw[32:36])[0]
        assert raw[36:36 + ln] == b"density"
        X, Y, Z = struct.unpack("<3Q", raw[36 + ln:60 + ln])
        assert (X, Y, Z) == data.shape
        ch, ty = struct.unpack("<2i", raw[60 + ln:68 + ln])
        assert (ch, ty) == (1, {np.uint8: 0, np.uint16: 1, np.float32: 2}[dtype])
        body = np.frombuffer(raw[68 + ln:], dtype).reshape(Z, Y, X)
        assert np.array_equal(body.transpose(2, 1, 0), data)
        vol = capi.Volume.load(path)
        res, bmin, bsize = vol.info()
        assert res == data.shape
        assert np.allclose(bmin, (-0.5, -0.4, -0.3)) and np.allclose(bsize, (1.0, 0.8, 0.6))


def _lz4_block_decode(src: bytes, out: bytearray) -> None:
    """the published LZ4 block format, decoded by a test-side decoder (independent of the library's)"""
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1; lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1; ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        assert 0 < off <= len(out)
        out += out[start:start + ml] if off >= ml else (bytes(out[start:start + off]) * ((ml + off - 1) // off))[:ml]


@pytest.mark.parametrize("compression", [1, 9])
def test_cvol_compressed_save(tmp_path, compression):
    """Volume::save(filename, compression > 0) (volume.cpp:623-682): Flag_Compressed, the body as LZ4 messages (int32 size + one block per 64 KiB).
    The file is decoded by the test's own LZ4 decoder (block format rules: the last sequence is literals only and at least five bytes long) and
    read back by the library; smooth, noisy, tiny (< 13 bytes: no match possible) and exactly-64-KiB inputs."""
    from fvsrn_amd import capi
    rng = np.random.RandomState(6)
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in (64, 40, 36)], indexing="ij")
    cases = dict(smooth=(np.round(np.exp(-3 * (x * x + y * y + z * z)) * 6) * 40).astype(np.uint8),
                 noisy16=rng.randint(0, 65536, (33, 20, 50)).astype(np.uint16),
                 float32=np.round(rng.rand(24, 20, 30) * 4).astype(np.float32) / 4,
                 tiny=rng.randint(0, 255, (3, 2, 2)).astype(np.uint8),
                 one_message=np.zeros((64, 32, 32), np.uint8),
                 constant=np.full((70, 40, 30), 7, np.uint8))
    for name, data in cases.items():
        path = str(tmp_path / (name + ".cvol"))
        capi.Volume.save_cvol(path, data, world_size=(1.0, 0.8, 0.6), feature_name="density", compression=compression)
        raw = open(path, "rb").read()
        assert raw[:4] == b"CVOL" and struct.unpack("<2i", raw[20:28]) == (1, 1), name  # one feature, Flag_Compressed
        ln = struct.unpack("<i", raw[32:36])[0]
        body, pos, out = raw[68 + ln:], 0, bytearray()
        want = np.ascontiguousarray(data.transpose(2, 1, 0)).tobytes()
        while len(out) < len(want):
            cs = struct.unpack("<i", body[pos:pos + 4])[0]; pos += 4
            before = len(out)
            block = body[pos:pos + cs]; pos += cs
            _lz4_block_decode(block, out)
            assert len(out) - before == min(65536, len(want) - before), name
            # end-of-block rule: the last token carries literals only (>= 5 of them unless the whole message is shorter)
            assert len(out) - before < 5 or bytes(out[-5:]) == bytes(block[-5:]), name
        assert pos == len(body) and bytes(out) == want, name
        if name in ("smooth", "one_message", "constant"):
            assert len(body) < len(want) // 4, (name, len(body), len(want))
        vol = capi.Volume.load(path)
        assert vol.info()[0] == data.shape
        scale = {np.dtype(np.uint8): 255.0, np.dtype(np.uint16): 65535.0, np.dtype(np.float32): 1.0}[data.dtype]
        assert np.array_equal(vol.data(), data.astype(np.float32) / np.float32(scale)), name
    with pytest.raises(capi.FvsrnError, match="compression"):
        capi.Volume.save_cvol(str(tmp_path / "bad.cvol"), cases["tiny"], compression=10)


def test_cvol_lz4_and_legacy_files(tmp_path):
    """LZ4-compressed bodies (Flag_Compressed, volume.cpp:647-664,721-735) in the lz4cpp framing -- int32 size + LZ4 block per 64 KiB message, ONE
    dependent-block stream per file, so a match in the second feature may reach into the first -- and the old density-only "cvol" format
    (volume.cpp:741-793), compressed and not.  The files are written by the test-side encoder (util.lz4_messages)."""
    from fvsrn_amd import capi
    rng = np.random.RandomState(3)
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in (48, 40, 36)], indexing="ij")
    a = (np.round(np.exp(-3 * (x * x + y * y + z * z)) * 6) * 40).astype(np.uint8)                           # plateaus: long matches
    b = rng.randint(0, 4, (40, 33, 50)).astype(np.uint16) * 1000                                             # noisy u16: short matches, > 64 KiB
    xfast = lambda v: np.ascontiguousarray(v.transpose(2, 1, 0)).tobytes()  # noqa: E731
    feat = lambda name, v, ty: struct.pack("<i", len(name)) + name + struct.pack("<3Q2i", *v.shape, 1, ty)  # noqa: E731
    # one stream per file: feature b's messages are encoded with feature a's bytes as their history (front padding puts b's first byte on a
    # message boundary of the encoder; distances between b and a are unchanged)
    path = tmp_path / "two.cvol"
    body_a = util.lz4_messages(xfast(a))
    hist = xfast(a)
    pad = (-len(hist)) % 65536
    cont = util.lz4_messages(bytes(pad) + hist + xfast(b))
    pos = 0
    for _ in range((pad + len(hist)) // 65536):
        pos += 4 + struct.unpack("<i", cont[pos:pos + 4])[0]
    body_b = cont[pos:]
    path.write_bytes(b"CVOL" + struct.pack("<i3f3i", 1, 1.0, 0.8, 0.6, 2, 1, 0) + feat(b"density", a, 0) + body_a + feat(b"other", b, 1) + body_b)
    va, vb = capi.Volume.load(str(path), 0), capi.Volume.load(str(path), 1)
    assert va.info()[0] == a.shape and vb.info()[0] == b.shape
    assert np.array_equal(va.data(), a.astype(np.float32) / 255.0)
    assert np.array_equal(vb.data(), b.astype(np.float32) / 65535.0)
    assert len(body_a) < a.size // 2, "the smooth volume should compress (matches are exercised)"
    for compressed in (0, 1):
        old = tmp_path / ("old%d.cvol" % compressed)
        old.write_bytes(b"cvol" + struct.pack("<3Q3dIB7x", *a.shape, 1 / 48, 1 / 48, 1 / 48, 0, compressed) + (util.lz4_messages(xfast(a)) if compressed else xfast(a)))
        v = capi.Volume.load(str(old))
        res, bmin, bsize = v.info()
        assert res == a.shape and np.allclose(bsize, (1.0, 40 / 48, 36 / 48)) and np.allclose(bmin, -bsize / 2)
        assert np.array_equal(v.data(), a.astype(np.float32) / 255.0)


REFERENCE_CVOL = "/root/reference/applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol"


@pytest.mark.skipif(not os.path.exists(REFERENCE_CVOL), reason="needs the reference checkout (build container only)")
def test_reference_volume_file_loads():
    """The one volume the reference snapshot holds (old format, 256^3 bytes, LZ4): what config-files/RichtmyerMeshkov-t20-v1-dvr.json points
    at.  Pinned by tests/golden/cvol_ppm_t0020.npz, which tests/golden/make_cvol_fixture.py wrote with an independent pure-Python decoder."""
    import zlib
    from fvsrn_amd import capi
    ref = np.load(os.path.join(util.GOLDEN_DIR, "cvol_ppm_t0020.npz"))
    v = capi.Volume.load(REFERENCE_CVOL)
    res, bmin, bsize = v.info()
    assert res == tuple(ref["resolution"]) == (256, 256, 256) and np.allclose(bsize, 1.0) and np.allclose(bmin, -0.5)
    d = np.rint(v.data() * 255.0).astype(np.uint8)
    assert (int(d.min()), int(d.max())) == (int(ref["minimum"]), int(ref["maximum"])) and abs(float(d.mean()) - float(ref["mean"])) < 1e-9
    assert np.array_equal(np.bincount(d.reshape(-1) >> 2, minlength=64), ref["histogram"])
    assert zlib.crc32(np.ascontiguousarray(d.transpose(2, 1, 0)).tobytes()) == int(ref["crc32"])
    assert np.abs(d.reshape(32, 8, 32, 8, 32, 8).astype(np.float64).mean(axis=(1, 3, 5)) / 255.0 - ref["block_mean_32"]).max() < 1e-6


def test_cvol_errors(tmp_path):
    from fvsrn_amd import capi
    bad = tmp_path / "bad.cvol"
    bad.write_bytes(b"XXXX" + b"\0" * 64)
    with pytest.raises(capi.FvsrnError, match="magic"):
        capi.Volume.load(str(bad))
    lz4 = tmp_path / "lz4.cvol"  # compressed flag, feature header, then a message that claims more bytes than the file has
    lz4.write_bytes(b"CVOL" + struct.pack("<i3f3i", 1, 1, 1, 1, 1, 1, 0) + struct.pack("<i", 1) + b"d" + struct.pack("<3Q2i", 4, 4, 4, 1, 0) + struct.pack("<i", 1000) + b"\x10a")
    with pytest.raises(capi.FvsrnError, match="LZ4"):
        capi.Volume.load(str(lz4))
    lz4.write_bytes(b"CVOL" + struct.pack("<i3f3i", 1, 1, 1, 1, 1, 1, 0) + struct.pack("<i", 1) + b"d" + struct.pack("<3Q2i", 4, 4, 4, 1, 0) + struct.pack("<i", 4) + b"\x0f\x01\x00\x05")
    with pytest.raises(capi.FvsrnError, match="LZ4"):  # a match that reaches in front of the stream
        capi.Volume.load(str(lz4))
    with pytest.raises(capi.FvsrnError, match="open"):
        capi.Volume.load(str(tmp_path / "missing.cvol"))


# ------------------------------------------------------------------------------------------------ GPU
def scene_kwargs(**kw):
    eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.5, 0.8, 1.7)
    d = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 96, early_out=True,
             tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=30.0, tf_scale_emission=1.0, density_min=0.1, density_max=0.9)
    d.update(kw)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("source", [oracle.VOLUME_SOURCE_TEXTURE, oracle.VOLUME_SOURCE_TENSOR])
@pytest.mark.parametrize("interp", [oracle.VOLUME_NEAREST, oracle.VOLUME_TRILINEAR, oracle.VOLUME_TRICUBIC])
@pytest.mark.parametrize("new_behavior", [False, True])
def test_evaluate_points_matches_restatement(source, interp, new_behavior):
    import torch
    from fvsrn_amd import capi
    data = make_volume(shape=(17, 13, 9))
    rng = np.random.RandomState(9)
    pos = (np.array(BOX_MIN) - 0.1 + rng.rand(4099, 3) * (np.array(BOX_SIZE) + 0.2)).astype(np.float32)  # also outside the box
    ref = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, interp, source, new_behavior).evaluate(pos)
    vol = capi.Volume.from_array(data, BOX_MIN, BOX_SIZE)
    got = vol.evaluate(torch.from_numpy(pos).cuda(), interp, source, new_behavior).cpu().numpy()[:, 0]
    # same fp32 operations; hipcc may contract a*b+c into fma where gcc does not: a few ulp
    assert np.abs(got - ref).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(dtype=np.float32, interp=oracle.VOLUME_TRILINEAR, source=oracle.VOLUME_SOURCE_TEXTURE, tf="identity"),
    dict(dtype=np.uint8, interp=oracle.VOLUME_TRILINEAR, source=oracle.VOLUME_SOURCE_TEXTURE, tf="texture"),
    dict(dtype=np.uint16, interp=oracle.VOLUME_TRICUBIC, source=oracle.VOLUME_SOURCE_TENSOR, tf="identity"),
    dict(dtype=np.float32, interp=oracle.VOLUME_NEAREST, source=oracle.VOLUME_SOURCE_TENSOR, tf="identity", new_behavior=True,
         early_out=False, blend_mode=oracle.BLEND_ALPHA),
])
def test_render_volume_matches_restatement(case):
    import torch
    from fvsrn_amd import capi
    data = make_volume(shape=(24, 20, 16), dtype=case["dtype"])
    kw = scene_kwargs(early_out=case.get("early_out", True), blend_mode=case.get("blend_mode", oracle.BLEND_BEER_LAMBERT))
    if case["tf"] == "texture":
        rng = np.random.RandomState(5)
        tab = rng.uniform(0.0, 1.0, (32, 4)).astype(np.float32)
        tab[:, 3] *= 40.0
        kw.update(tf_kind=oracle.TF_TEXTURE, tf_table=tab)
    W, H = 72, 56  # not multiples of the 16x16 pixel blocks
    nb = case.get("new_behavior", False)
    ref, count = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, case["interp"], case["source"], nb).render(oracle.OracleScene(**kw), W, H)
    vol = capi.Volume.from_array(data, BOX_MIN, BOX_SIZE)
    stats = torch.zeros(2, dtype=torch.int64, device="cuda")
    img = vol.render(capi.Scene(**kw), W, H, case["interp"], case["source"], nb, stats=stats)[0].cpu().numpy()
    assert ref[3].max() > 0.5
    assert np.array_equal(np.isnan(img[7]), np.isnan(ref[7]))
    # nearest-neighbour lookups flip voxels where a coordinate rounds differently by one ulp: compare all but a few pixels there
    diff = np.abs(img[:4] - ref[:4]).max(axis=0)
    if case["interp"] == oracle.VOLUME_NEAREST:
        assert (diff > 1e-4).mean() < 0.01
    elif case["source"] == oracle.VOLUME_SOURCE_TEXTURE:
        # the 8-bit filter weights of the texture model jump by 1/256 where a coordinate differs by one ulp (the compilers
        # contract eye + dir * t differently): a random 32-entry TF turns that into <= 1e-3 of colour
        assert diff.max() < 3e-3 and (diff > 1e-4).mean() < 0.02
    else:
        assert diff.max() < 1e-4
    st = stats.cpu().numpy()
    if case.get("early_out", True):  # depth segments (small images) stop early per segment: a few more samples than the single loop
        assert count <= int(st[0]) <= 1.5 * count
    else:
        assert abs(int(st[0]) - count) <= max(8, count // 10000)


PHONG = dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8,
             light_type=1, light=(0.3, 0.5, -1.0), enable_magnitude_scaling=True, magnitude_scaling=3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(brdf=None, normals=True, preint=0),      # normal channel only (ChannelNormal)
    dict(brdf=PHONG, normals=False, preint=0),    # Phong + magnitude scaling from the grid's central-difference gradients
    # pre-integrated Texture TF, 2D table (looks at the previous sample of the ray; the 1D mode switches formulas at
    # |d - d_prev| = 1e-3, where one ulp of density flips isolated pixels by a few percent)
    dict(brdf=None, normals=False, preint=2),
    # Gaussian TF variants (renderer_tf_gaussian.cuh:55-73): sigma scaled by the grid's gradient length (the TF requests normals,
    # transfer_function_gaussian.cpp:271-272) / closed-form integration between consecutive samples
    dict(brdf=None, normals=False, preint=0, gauss=1),
    dict(brdf=None, normals=False, preint=0, gauss=2),
])
def test_render_volume_normals_shading_preintegration(case):
    """evalNormalImpl (renderer_volume_grid.cuh:234-283) feeding BRDFLambert and the normal channels; pre-integrated TFs."""
    import torch
    from fvsrn_amd import capi
    data = make_volume(shape=(24, 20, 16))
    kw = scene_kwargs(brdf=case["brdf"])
    if case["preint"]:
        rng = np.random.RandomState(5)
        tab = rng.uniform(0.0, 1.0, (64, 4)).astype(np.float32)
        tab[:, 3] *= 40.0
        kw.update(tf_kind=oracle.TF_TEXTURE, tf_table=tab, tf_preintegration=case["preint"])
    if case.get("gauss"):
        tab = np.array([[0.9, 0.1, 0.1, 30.0, 0.25, 0.08], [0.1, 0.9, 0.2, 60.0, 0.5, 0.05], [0.2, 0.3, 0.95, 90.0, 0.8, 0.1]], np.float32)
        if case["gauss"] == 1:
            tab[:, 5] *= 4.0  # gradients of this volume are a few per unit length: sigma * 0.1 |g|
        kw.update(tf_kind=oracle.TF_GAUSSIAN, tf_table=tab, tf_gaussian_mode=case["gauss"])
    W, H = 64, 48
    ov = oracle.OracleVolume(data, BOX_MIN, BOX_SIZE, oracle.VOLUME_TRILINEAR, oracle.VOLUME_SOURCE_TENSOR, provide_normals=case["normals"])
    ref, _ = ov.render(oracle.OracleScene(**kw), W, H)
    vol = capi.Volume.from_array(data, BOX_MIN, BOX_SIZE)
    img = vol.render(capi.Scene(**kw), W, H, oracle.VOLUME_TRILINEAR, oracle.VOLUME_SOURCE_TENSOR, provide_normals=case["normals"])[0].cpu().numpy()
    assert ref[3].max() > (0.05 if case.get("gauss") else 0.5)
    # Phong: rsqrt / powf / exp of the device vs libm on gradients of ~10/unit: looser, like the shaded network renders
    assert np.abs(img[:4] - ref[:4]).max() < (5e-3 if case["brdf"] or case.get("gauss") else 5e-4)
    if case.get("gauss"):  # (the gradient-scaled variant makes the volume provide normals: they show up in the normal channels)
        plain, _ = ov.render(oracle.OracleScene(**dict(kw, tf_gaussian_mode=0)), W, H)
        assert np.abs(plain[:4] - ref[:4]).max() > 2e-2
        assert np.abs(img[4:7] - ref[4:7]).max() < 2e-3 and (np.abs(ref[4:7]).max() > 0.1) == (case["gauss"] == 1)
    elif case["normals"] or case["brdf"]:
        assert np.abs(ref[4:7]).max() > 0.1 and np.abs(img[4:7] - ref[4:7]).max() < (5e-3 if case["brdf"] else 2e-3)
    else:
        assert np.abs(img[4:7]).max() == 0.0
