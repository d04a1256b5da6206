"""Host-side mirror of the reference interface (no GPU): transformations, HdrImage / PFM,
tone mapping, flattening to the C-ABI tables, row-tile partition bookkeeping.
Each test names the reference #[test] it restates (paths relative to /root/reference)."""
import ctypes as C
import io
import math
import os
import re

import numpy as np
import pytest

import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.hdrimage import _parse_endianness, _parse_img_shape
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.transformation import _matmul

PI = np.float32(math.pi)
M1 = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 8, 7], [6, 5, 4, 1]]
M1I = [[-3.75, 2.75, -1, 0], [4.375, -3.875, 2, -0.5], [0.5, 0.5, -1, 1], [-1.375, 0.875, 0, -0.5]]


# ---- src/transformation.rs:382-652 -----------------------------------------------------------
def test_mul_matrix_exact():
    m1 = np.array([[1, 2, -1, 3], [1, -2, 0, 2], [1, -1, 0, 1], [0, 3, -3, 1]], dtype=np.float32)
    m2 = np.array([[-2, 2, 1, 3], [1, -1, 0, 1], [1, 2, -1, 1], [1, -2, 0, -3]], dtype=np.float32)
    exp = np.array([[2, -8, 2, -5], [-2, 0, 1, -5], [-2, 1, 1, -1], [1, -11, 3, -3]], dtype=np.float32)
    assert np.array_equal(_matmul(m1, m2), exp)  # assert_eq!


def test_is_consistent_is_close_inverse():
    assert ra.Transformation().is_consistent()
    m1 = ra.Transformation(M1, M1I)
    assert m1.is_consistent() and m1.is_close(m1)
    m2 = ra.Transformation(M1, M1I); m2.m[2, 2] += 1.0
    m3 = ra.Transformation(M1, M1I); m3.invm[2, 2] += 1.0
    assert not m1.is_close(m2) and not m1.is_close(m3)
    inv = m1.inverse()
    assert inv.is_consistent() and (m1 * inv).is_close(ra.Transformation())


def test_mul_transformation():
    m1 = ra.Transformation(M1, M1I)
    m2 = ra.Transformation([[3, 5, 2, 4], [4, 1, 0, 5], [6, 3, 2, 0], [1, 4, 2, 1]],
                           [[0.4, -0.2, 0.2, -0.6], [2.9, -1.7, 0.2, -3.1], [-5.55, 3.15, -0.4, 6.45], [-0.9, 0.7, -0.2, 1.1]])
    exp = ra.Transformation([[33, 32, 16, 18], [89, 84, 40, 58], [118, 106, 48, 88], [63, 51, 22, 50]],
                            [[-1.45, 1.45, -1, 0.6], [-13.95, 11.95, -6.5, 2.6], [25.525, -22.025, 12.25, -5.2], [4.825, -4.325, 2.5, -1.1]])
    assert m2.is_consistent() and exp.is_consistent() and exp.is_close(m1 * m2)


def test_translation_scaling_rotations():
    assert (ra.translation((1, 2, 3)) * ra.translation((4, 6, 8))).is_close(ra.translation((5, 8, 11)))
    assert (ra.scaling((2, 5, 10)) * ra.scaling((3, 2, 4))).is_close(ra.scaling((6, 10, 40)))
    for t in (ra.rotation_x(12.3), ra.rotation_y(-0.14), ra.rotation_z(1.0), ra.translation((1, 2, 3)), ra.scaling((2, 5, 10))):
        assert t.is_consistent()
    eye = np.eye(4, dtype=np.float32)
    assert np.all(np.abs(ra.rotation_x(np.float32(2) * PI).m - eye) < 1e-4)
    assert np.all(np.abs(ra.rotation_y(np.float32(-4) * PI).invm - eye) < 1e-4)
    assert np.all(np.abs(ra.rotation_z(0.0).m - eye) < 1e-4)


def test_mul_vector_normal_point_ray():
    close = lambda a, b: all(abs(float(x) - float(y)) < 1e-4 for x, y in zip(a, b))
    assert close(ra.rotation_x(PI / np.float32(3)).apply_vector((1, 1, 0)), (1, 0.5, math.sqrt(3) / 2))
    assert close(ra.scaling((2, -3, 5)).apply_normal((2, 1, 0)), (1, -1 / 3, 0))
    assert close(ra.translation((1, -2, 3)).apply_point((-3, 2, 0)), (-2, 0, 3))
    m = ra.Transformation([[1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 8, 7], [0, 0, 0, 1]],
                          [[-3.75, 2.75, -1, 0], [5.75, -4.75, 2, 1], [-2.25, 2.25, -1, -2], [0, 0, 0, 1]])
    assert m.is_consistent()
    assert close(m.apply_vector((1, 2, 3)), (14, 38, 51)) and close(m.apply_point((1, 2, 3)), (18, 46, 58))
    assert close(m.apply_normal((3, 2, 4)), (-8.75, 7.75, -3))
    t = ra.translation((10, 11, 12)) * ra.rotation_x(PI / np.float32(2))
    assert close(t.apply_point((1, 2, 3)), (11, 8, 14)) and close(t.apply_vector((6, 5, 4)), (6, -4, 5))


def test_to_radians_is_f32():
    assert ra.to_radians(230.0) == np.float32(230.0) * (np.float32(math.pi) / np.float32(180.0))


# ---- src/hdrimage.rs:430-868 -----------------------------------------------------------------
def _ref_image():
    img = ra.HdrImage(3, 2)
    vals = [(1e1, 2e1, 3e1), (4e1, 5e1, 6e1), (7e1, 8e1, 9e1), (1e2, 2e2, 3e2), (4e2, 5e2, 6e2), (7e2, 8e2, 9e2)]
    for i, c in enumerate(vals):
        img.set_pixel(i % 3, i // 3, c)
    return img


REF_BE = bytes([0x50, 0x46, 0x0a, 0x33, 0x20, 0x32, 0x0a, 0x31, 0x2e, 0x30, 0x0a, 0x42, 0xc8, 0x00, 0x00, 0x43, 0x48, 0x00,
                0x00, 0x43, 0x96, 0x00, 0x00, 0x43, 0xc8, 0x00, 0x00, 0x43, 0xfa, 0x00, 0x00, 0x44, 0x16, 0x00, 0x00, 0x44,
                0x2f, 0x00, 0x00, 0x44, 0x48, 0x00, 0x00, 0x44, 0x61, 0x00, 0x00, 0x41, 0x20, 0x00, 0x00, 0x41, 0xa0, 0x00,
                0x00, 0x41, 0xf0, 0x00, 0x00, 0x42, 0x20, 0x00, 0x00, 0x42, 0x48, 0x00, 0x00, 0x42, 0x70, 0x00, 0x00, 0x42,
                0x8c, 0x00, 0x00, 0x42, 0xa0, 0x00, 0x00, 0x42, 0xb4, 0x00, 0x00])
REF_LE = bytes([0x50, 0x46, 0x0a, 0x33, 0x20, 0x32, 0x0a, 0x2d, 0x31, 0x2e, 0x30, 0x0a, 0x00, 0x00, 0xc8, 0x42, 0x00, 0x00,
                0x48, 0x43, 0x00, 0x00, 0x96, 0x43, 0x00, 0x00, 0xc8, 0x43, 0x00, 0x00, 0xfa, 0x43, 0x00, 0x00, 0x16, 0x44,
                0x00, 0x00, 0x2f, 0x44, 0x00, 0x00, 0x48, 0x44, 0x00, 0x00, 0x61, 0x44, 0x00, 0x00, 0x20, 0x41, 0x00, 0x00,
                0xa0, 0x41, 0x00, 0x00, 0xf0, 0x41, 0x00, 0x00, 0x20, 0x42, 0x00, 0x00, 0x48, 0x42, 0x00, 0x00, 0x70, 0x42,
                0x00, 0x00, 0x8c, 0x42, 0x00, 0x00, 0xa0, 0x42, 0x00, 0x00, 0xb4, 0x42])


def test_image_basics():
    img = ra.HdrImage(5, 7)
    assert img.shape() == (5, 7)
    assert img.valid_coordinates(0, 0) and img.valid_coordinates(4, 3)
    assert not img.valid_coordinates(7, 0) and not img.valid_coordinates(0, 9) and not img.valid_coordinates(7, 9)
    img = ra.HdrImage(3, 3)
    assert img.get_pixel(0, 0) == (0, 0, 0)
    with pytest.raises(ra.HdrImageErr):
        img.get_pixel(3, 3)
    img.set_pixel(2, 2, (1.23, 4.56, 7.89))
    assert np.allclose(img.get_pixel(2, 2), (1.23, 4.56, 7.89))
    with pytest.raises(ra.HdrImageErr):
        img.set_pixel(5, 5, (1, 1, 1))
    img.set_pixels([ra.WHITE] * 9)
    assert img.get_pixel(2, 2) == (1, 1, 1)
    with pytest.raises(ra.HdrImageErr):
        img.set_pixels([ra.BLACK] * 11)


def test_parse_img_shape_and_endianness():
    assert _parse_img_shape("10 20") == (10, 20) and _parse_img_shape(" 10    20  ") == (10, 20)
    for bad in ("10 20 30", "10 ", "102030", "10 20.1", "-10 20", "abc 20"):
        with pytest.raises(ra.HdrImageErr):
            _parse_img_shape(bad)
    assert _parse_endianness("-3.2") == "little" and _parse_endianness("1e15") == "big"
    for bad in ("0", "abc"):
        with pytest.raises(ra.HdrImageErr):
            _parse_endianness(bad)


def test_read_pfm_image():
    ref = _ref_image()
    for data in (REF_BE, REF_LE):
        img = ra.HdrImage.read_pfm_image(io.BytesIO(data))
        assert img.shape() == (3, 2) and np.array_equal(img.pixels, ref.pixels)
    for bad in (bytes([0x46, 0x50, 0x0a]), bytes([0xff, 0xff, 0x0a]), bytes([0x50, 0x46]),
                bytes([0x50, 0x46, 0x0a, 0xff, 0xff, 0x0a]), REF_LE[:11], REF_LE + b"\x00", REF_LE[:-1]):
        with pytest.raises(ra.HdrImageErr):
            ra.HdrImage.read_pfm_image(io.BytesIO(bad))


def test_write_pfm_image_and_file(tmp_path):
    img = _ref_image()
    for endian, ref in (("big", REF_BE), ("little", REF_LE)):
        buf = io.BytesIO()
        img.write_pfm_image(buf, endian)
        assert buf.getvalue() == ref  # byte-exact
        path = tmp_path / f"ref_{endian}.pfm"
        img.write_pfm_file(path, endian)
        assert np.array_equal(ra.HdrImage.read_pfm_file(path).pixels, img.pixels)
    with pytest.raises(OSError):
        img.write_pfm_file("/invalid_dir_xyz/file.pfm")


def test_tone_mapping(tmp_path):
    img = ra.HdrImage(2, 1)
    img.set_pixel(0, 0, (5, 10, 15)); img.set_pixel(1, 0, (500, 1000, 1500))
    # hdrimage.rs:779-791 at f32 exactness: log10f(10) = 1, log10f(1000) = 3, powf(10, 2) = 100
    assert img.average_luminosity() == np.float32(100.0) and img.average_luminosity().dtype == np.float32
    img.normalize_image(1000.0, 100.0)
    assert np.allclose(img.get_pixel(0, 0), (0.5e2, 1e2, 1.5e2)) and np.allclose(img.get_pixel(1, 0), (0.5e4, 1e4, 1.5e4))
    img.clamp_image()
    assert np.all((img.pixels >= 0) & (img.pixels < 1))
    img = ra.HdrImage.read_pfm_image(io.BytesIO(REF_LE))
    img.normalize_image(1000.0); img.clamp_image()
    img.write_ldr_file(tmp_path / "ref.png", 1.0)
    img.write_ldr_file(tmp_path / "ref.ff", 1.0)
    with pytest.raises(ra.HdrImageErr):
        img.write_ldr_file(tmp_path / "ref.jpeg", 1.0)
    from PIL import Image
    png = np.asarray(Image.open(tmp_path / "ref.png"))
    assert png.shape == (2, 3, 3) and np.array_equal(png, img.ldr_bytes(1.0, 8))
    ff = (tmp_path / "ref.ff").read_bytes()
    assert ff[:8] == b"farbfeld" and len(ff) == 16 + 3 * 2 * 8


# ---- flattening to the C-ABI tables ------------------------------------------------------------
def test_flatten_demo_world_preserves_order_and_dedups_materials():
    flat = demo_world().flatten()
    assert (flat.n_shapes, flat.n_materials, flat.n_textures) == (4, 4, 0)
    assert [flat.shapes[i].kind for i in range(4)] == [0, 1, 0, 0]
    assert list(flat.shapes[0].m)[:4] == [200.0, 0.0, 0.0, 0.0] and abs(flat.shapes[0].m[11] - 0.4) < 1e-7
    assert flat.materials[flat.shapes[3].material].brdf == _abi.RT_BRDF_SPECULAR
    assert flat.materials[flat.shapes[1].material].brdf_pigment.kind == _abi.RT_PIGMENT_CHECKERED
    shared = ra.Material()
    w = ra.World(); w.add(ra.Sphere(material=shared)); w.add(ra.Plane(material=shared))
    assert w.flatten().n_materials == 1
    cam = ra.flatten_camera(demo_camera(640, 480))
    assert cam.kind == _abi.RT_CAMERA_PERSPECTIVE and abs(cam.aspect - 640 / 480) < 1e-6


def test_abi_struct_sizes_match_header():
    sizes = {_abi.rt_shape: 136, _abi.rt_pigment: 36, _abi.rt_material: 76, _abi.rt_texture: 16, _abi.rt_camera: 76,
             _abi.rt_renderer: 40, _abi.rt_ray: 36, _abi.rt_hit: 44, _abi.rt_pcg: 16, _abi.rt_partition: 20,
             _abi.rt_stats: 72}
    for t, n in sizes.items():
        assert C.sizeof(t) == n, t


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/rustracer_amd.h declares."""
    header = open(os.path.join(os.path.dirname(_abi.LIB_PATH), "..", "include", "rustracer_amd.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header))
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    lib = _abi.lib()
    for name in declared:
        assert hasattr(lib, name)


def test_partition_rows_bookkeeping():
    from bench import partition_rows
    lib = _abi.lib()
    for h, n_parts, tile in [(1080, 8, 1), (1080, 8, 4), (90, 4, 100), (7, 3, 2), (1, 2, 1)]:
        rows = [partition_rows(h, p, n_parts, tile) for p in range(n_parts)]
        assert sorted(r for part in rows for r in part) == list(range(h))
        for p in range(n_parts):
            assert lib.rt_partition_pixels(13, h, C.byref(_abi.rt_partition(p, n_parts, tile))) == 13 * len(rows[p])


def test_tone_mapping_f32_details():
    """Color::luminosity ignores a NaN channel (f32::max / f32::min, src/color.rs:33-36); the extension match is
    case-insensitive (ImageFormat::from_path); --output-pfm strips one extension and replaces the next (src/main.rs:225)."""
    from rustracer_amd.cli import pfm_path_for
    img = ra.HdrImage(2, 1)
    img.set_pixel(0, 0, (5, float("nan"), 15)); img.set_pixel(1, 0, (500, 1000, 1500))
    assert img.average_luminosity() == np.float32(100.0)          # max/min of (5, NaN, 15) = 15 / 5, as f32::max does
    # sequential f32 accumulation, not pairwise: 2^24 + 1 + 1 ... stays 2^24 in f32
    big = ra.HdrImage(5, 1)
    lum = np.array([10.0 ** 7.5] + [10.0 ** 1e-7] * 4, dtype=np.float32)
    big.set_pixels(np.repeat(lum[:, None], 3, axis=1).reshape(1, 5, 3))
    logs = np.log10((np.float32(1e-10) + lum).astype(np.float32))
    seq = np.float32(0.0)
    for v in logs:
        seq = np.float32(seq + v)
    assert big.average_luminosity() == np.float32(np.power(np.float32(10.0), np.float32(seq / np.float32(5))))
    assert pfm_path_for("x.y.png") == "x.pfm" and pfm_path_for("out.png") == "out.pfm" and pfm_path_for("d/a.ff") == "d/a.pfm"
    img.normalize_image(1.0, 100.0); img.clamp_image()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        img.write_ldr_file(os.path.join(d, "UPPER.PNG"))
        assert os.path.getsize(os.path.join(d, "UPPER.PNG")) > 0
