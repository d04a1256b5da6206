"""Pins oracle/rt_oracle.cpp against the reference's own known-answer tests.

Each test names the reference #[test] it restates (paths relative to /root/reference).  Tolerance
is the reference's IsClose (abs 1e-4, src/misc.rs:6-23) unless the reference uses assert_eq!.
CPU only.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.world import flatten_camera, flatten_world, to_rt_ray

EPS = 1e-4
INF = math.inf


def close(a, b, eps=EPS):
    return all(abs(float(x) - float(y)) < eps for x, y in zip(a, b))


def ray(origin=(0, 0, 0), dir=(1, 0, 0), **kw):
    return to_rt_ray(ra.Ray(origin=origin, dir=dir, **kw))


def world_of(*shapes):
    w = ra.World()
    for s in shapes:
        w.add(s)
    return w


def one_hit(world, r, mode=O.MATH_LIBM):
    return O.world_ray_intersection(flatten_world(world), [r], mode)[0]


def assert_hit(h, world_point, normal, uv, t):
    assert h.hit == 1
    assert close(h.world_point, world_point) and close(h.normal, normal) and close(h.uv, uv)
    assert abs(h.t - t) < EPS


# ---- src/random.rs:56-81 -------------------------------------------------------------------
def test_random():
    p = _abi.rt_pcg()
    O.lib().oracle_pcg_new(42, 54, C.byref(p))
    assert p.state == 1753877967969059832 and p.inc == 109
    s = np.array([42], dtype=np.uint64)
    q = np.array([54], dtype=np.uint64)
    out = np.zeros(6, dtype=np.uint32)
    O.lib().oracle_pcg_random(s.ctypes.data_as(C.POINTER(C.c_uint64)), q.ctypes.data_as(C.POINTER(C.c_uint64)), 1, 6,
                              out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert out.tolist() == [2707161783, 2068313097, 3122475824, 2211639955, 3215226955, 3421331566]


def test_random_float():
    pcg = O.OraclePcg(38, 62)
    expected = np.array([0.090021014, 0.39037937, 0.6641162, 0.42459878, 0.30006477, 0.1585743], dtype=np.float32)
    got = pcg.random_floats(6)
    assert got.tobytes() == expected.tobytes()  # assert_eq! on f32


# ---- src/imagetracer.rs:61-84, derived check values of SURVEY.md §8 a2 -----------------------
def test_all_rays_seeds():
    w, h = 4, 2
    uv = np.zeros(w * h * 2, dtype=np.float32)
    seq = np.zeros(w * h, dtype=np.uint64)
    O.lib().oracle_all_rays_seeds(w, h, 42, 54, 1, O.fptr(uv), seq.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert (np.float32(0.63031024), np.float32(0.48156667), 3122475824) == (uv[0], uv[1], int(seq[0]))
    assert (np.float32(0.5149376), np.float32(0.74860334), 3421331566) == (uv[2], uv[3], int(seq[1]))
    assert (np.float32(0.74912477), np.float32(0.5046386), 3860803674) == (uv[4], uv[5], int(seq[2]))
    # the draws are the Pcg::default() stream: 2 jitter draws + 1 seq per pixel (random.rs:61)
    assert int(seq[0]) == 3122475824 and int(seq[1]) == 3421331566


# ---- src/shape.rs:223-568 ----------------------------------------------------------------
def test_hit_sphere():
    w = world_of(ra.Sphere())
    assert one_hit(w, ray((0, 10, 2), (0, 0, -1))).hit == 0
    assert_hit(one_hit(w, ray((0, 0, 2), (0, 0, -1))), (0, 0, 1), (0, 0, 1), (0, 0), 1.0)
    assert_hit(one_hit(w, ray((3, 0, 0), (-1, 0, 0))), (1, 0, 0), (1, 0, 0), (0, 0.5), 2.0)
    assert_hit(one_hit(w, ray((0, 0, 0), (1, 0, 0))), (1, 0, 0), (-1, 0, 0), (0, 0.5), 1.0)


def test_transform_sphere():
    w = world_of(ra.Sphere(ra.translation((10, 0, 0))))
    assert_hit(one_hit(w, ray((10, 0, 2), (0, 0, -1))), (10, 0, 1), (0, 0, 1), (0, 0), 1.0)
    assert_hit(one_hit(w, ray((13, 0, 0), (-1, 0, 0))), (11, 0, 0), (1, 0, 0), (0, 0.5), 2.0)
    assert one_hit(w, ray((0, 0, 2), (0, 0, -1))).hit == 0
    assert one_hit(w, ray((-10, 0, 0), (0, 0, -1))).hit == 0


def _normalized(v):
    n = math.sqrt(sum(float(c) ** 2 for c in v))
    return [float(c) / n for c in v]


def test_sphere_normal():
    h1 = one_hit(world_of(ra.Sphere(ra.scaling((2, 1, 1)))), ray((1, 1, 0), (-1, -1, 0)))
    assert h1.hit and close(_normalized(h1.normal), _normalized((1, 4, 0)))
    h2 = one_hit(world_of(ra.Sphere(ra.scaling((-1, -1, -1)))), ray((0, 2, 0), (0, -1, 0)))
    assert h2.hit and close(_normalized(h2.normal), _normalized((0, 1, 0)))


@pytest.mark.parametrize("mode", [O.MATH_LIBM, O.MATH_STRICT])
def test_sphere_point_to_uv(mode):
    w = world_of(ra.Sphere())
    cases = [((2, 0, 0), (-1, 0, 0), (0, 0.5)), ((0, 2, 0), (0, -1, 0), (0.25, 0.5)),
             ((-2, 0, 0), (1, 0, 0), (0.5, 0.5)), ((0, -2, 0), (0, 1, 0), (0.75, 0.5)),
             ((2, 0, 0.5), (-1, 0, 0), (0, 1 / 3)), ((2, 0, -0.5), (-1, 0, 0), (0, 2 / 3))]
    for o, d, uv in cases:
        assert close(one_hit(w, ray(o, d), mode).uv, uv)


def test_hit_plane():
    w = world_of(ra.Plane())
    assert_hit(one_hit(w, ray((0, 0, 1), (0, 0, -1))), (0, 0, 0), (0, 0, 1), (0, 0), 1.0)
    for d in [(0, 0, 1), (1, 0, 0), (0, 1, 0)]:
        assert one_hit(w, ray((0, 0, 1), d)).hit == 0


def test_transform_plane():
    w = world_of(ra.Plane(ra.rotation_y(np.float32(math.pi) / np.float32(2.0))))
    assert_hit(one_hit(w, ray((1, 0, 0), (-1, 0, 0))), (0, 0, 0), (1, 0, 0), (0, 0), 1.0)
    for d in [(0, 0, 1), (1, 0, 0), (0, 1, 0)]:
        assert one_hit(w, ray((0, 0, 1), d)).hit == 0


def test_plane_point_to_uv():
    w = world_of(ra.Plane())
    for o, uv in [((0, 0, 1), (0, 0)), ((0.25, 0.75, 1), (0.25, 0.75)), ((4.25, 7.75, 1), (0.25, 0.75)),
                  ((-4.25, -7.75, 1), (0.75, 0.25))]:
        assert close(one_hit(w, ray(o, (0, 0, -1))).uv, uv)


# ---- src/world.rs:62-92 ------------------------------------------------------------------
def test_world():
    white_emitter = ra.Material(ra.DiffuseBRDF(), ra.UniformPigment(ra.WHITE))
    w = world_of(ra.Sphere(), ra.Sphere(ra.translation((4, 0, 0)) * ra.scaling((2, 2, 2)), white_emitter))
    assert one_hit(w, ray((-2, 3, 0))).hit == 0
    h2 = one_hit(w, ray((-2, 0, 0)))
    assert h2.hit and h2.shape == 0  # default material: emitted BLACK
    h3 = one_hit(w, ray((-2, 1.5, 0)))
    assert h3.hit and h3.shape == 1  # emitted WHITE


def test_world_tie_break_first_shape_wins():
    # strict `<` at world.rs:38: two identical spheres -> index 0
    w = world_of(ra.Sphere(), ra.Sphere())
    assert one_hit(w, ray((-2, 0, 0))).shape == 0


# ---- src/render.rs:221-313 ---------------------------------------------------------------
def _solve(world, renderer, r, pcg, mode=O.MATH_LIBM):
    cols, new = O.renderer_solve(flatten_world(world), renderer.flatten(), [r], [(pcg.state, pcg.inc)], mode)
    pcg.state, pcg.inc = new[0]
    return cols[0]


def test_flat():
    red, green, blue = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    mat = ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment(red, blue, 2)), ra.UniformPigment(green))
    w = world_of(ra.Sphere(ra.Transformation(), mat))
    flat = ra.FlatRenderer(w, ra.BLACK)
    pcg = O.OraclePcg()
    assert close(_solve(w, flat, ray((-2, 3, 0)), pcg), (0, 0, 0))
    assert close(_solve(w, flat, ray((-2, 0.5, 0.5)), pcg), (1, 1, 0))
    assert close(_solve(w, flat, ray((-2, -0.5, 0.5)), pcg), (0, 1, 1))


def test_onoff():
    w = world_of(ra.Sphere())
    onoff = ra.OnOffRenderer(w, ra.BLACK, ra.WHITE)
    pcg = O.OraclePcg()
    assert close(_solve(w, onoff, ray((-2, 3, 0)), pcg), (0, 0, 0))
    assert close(_solve(w, onoff, ray((-2, 0, 0)), pcg), (1, 1, 1))


@pytest.mark.parametrize("mode", [O.MATH_LIBM, O.MATH_STRICT])
def test_furnace(mode):
    pcg = O.OraclePcg()
    for _ in range(10):
        emitted = pcg.random_float()
        reflectance = float(np.float32(pcg.random_float()) * np.float32(0.9))
        mat = ra.Material(ra.DiffuseBRDF(ra.UniformPigment((reflectance,) * 3)), ra.UniformPigment((emitted,) * 3))
        w = world_of(ra.Sphere(ra.Transformation(), mat))
        pt = ra.PathTracer(w, ra.BLACK, 1, 100, 101)
        color = _solve(w, pt, ray(), pcg, mode)
        expected = emitted / (1.0 - reflectance)
        assert close(color, (expected,) * 3)


def test_background():
    w = world_of(ra.Sphere(ra.translation((2, 0, 0))))
    pt = ra.PathTracer(w, ra.BLACK, 1000, 1000, 0)
    # NOTE the reference ray starts at the origin pointing +x and hits the sphere at x=1..3? No:
    # translation(E1*2) puts the unit sphere at x in [1,3]; Ray::default() origin (0,0,0) dir +x hits
    # it, the default material is WHITE diffuse / BLACK emitter and rr_limit = 0 -> black in the end
    # only through roulette.  The reference asserts is_close(BLACK).
    pcg = O.OraclePcg()
    assert close(_solve(w, pt, ray(), pcg), (0, 0, 0))


# ---- src/material.rs:324-405 ---------------------------------------------------------------
def _pigment_color(pig, uv):
    textures, tex_index, keep = [], {}, []
    from rustracer_amd.world import _fill_pigment
    p = _abi.rt_pigment()
    _fill_pigment(p, pig, textures, tex_index, keep)
    tex = (_abi.rt_texture * max(1, len(textures)))(*textures)
    uvs = np.array(uv, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    assert O.lib().oracle_pigment_get_color(C.byref(p), tex, len(textures), O.fptr(uvs), 1, O.fptr(out)) == 0
    return tuple(out.tolist())


def test_pigment():
    checkered = ra.CheckeredPigment(ra.BLACK, ra.WHITE, 10)
    img = ra.HdrImage(3, 3)
    img.set_pixel(0, 2, ra.WHITE)
    img.set_pixel(2, 0, ra.WHITE)
    img.set_pixel(2, 2, ra.WHITE)
    image = ra.ImagePigment(img)
    assert _pigment_color(ra.UniformPigment(), (0.1, 3.0)) == ra.BLACK
    assert _pigment_color(ra.UniformPigment(ra.WHITE), (0.5, 0.3)) == ra.WHITE
    assert _pigment_color(checkered, (0.0, 0.0)) == ra.BLACK
    assert _pigment_color(checkered, (2.0, 2.0)) == ra.BLACK
    assert _pigment_color(checkered, (0.0, 0.9)) == ra.WHITE
    assert _pigment_color(checkered, (0.9, 0.0)) == ra.WHITE
    assert _pigment_color(image, (0.0, 0.0)) == ra.BLACK
    assert _pigment_color(image, (0.0, 1.0)) == ra.WHITE
    assert _pigment_color(image, (1.0, 0.0)) == ra.WHITE
    assert _pigment_color(image, (1.0, 1.0)) == ra.WHITE
    # Rust float->u32 casts saturate, NaN -> 0
    assert _pigment_color(image, (float("nan"), float("nan"))) == ra.BLACK
    assert _pigment_color(image, (-5.0, 1e20)) == ra.WHITE  # col 0, row 2


def _scatter(brdf, r0, r1, incoming, point, normal, depth, mode=O.MATH_LIBM):
    out = _abi.rt_ray()
    f3 = lambda v: (C.c_float * 3)(*[float(c) for c in v])
    O.lib().oracle_scatter_ray(brdf, r0, r1, f3(incoming), f3(point), f3(normal), depth, mode, C.byref(out))
    return out


@pytest.mark.parametrize("mode", [O.MATH_LIBM, O.MATH_STRICT])
def test_brdf(mode):
    pcg = O.OraclePcg()
    r = pcg.random_floats(2)
    spec = _scatter(_abi.RT_BRDF_SPECULAR, r[0], r[1], (-1, 0, -1), (0, 0, 0), (0, 0, 1), 10, mode)
    inv = np.float32(1.0) / np.sqrt(np.float32(2.0))
    expected = np.array([-1, 0, 1], dtype=np.float32) * inv  # Vector::normalize (vector.rs:72-75)
    assert np.array(list(spec.dir), dtype=np.float32).tobytes() == expected.tobytes()  # assert_eq!
    assert spec.depth == 10 and spec.tmin == np.float32(1e-5) and spec.tmax == INF
    # mean of cosine-weighted directions ~ (0, 0, 2/3)
    num = 100_000  # reference uses 5e5; the bound scales as 1/sqrt(num)
    rf = pcg.random_floats(2 * num)
    acc = np.zeros(3, dtype=np.float64)
    for i in range(num):
        d = np.array(list(_scatter(_abi.RT_BRDF_DIFFUSE, rf[2 * i], rf[2 * i + 1], (0, 1, 0), (0, 0, 0),
                                   (0, 0, 1), 10, mode).dir), dtype=np.float64)
        acc += d / np.linalg.norm(d)
    acc /= num
    eps = 3.0 / math.sqrt(num)
    assert abs(acc[0]) <= eps and abs(acc[1]) <= eps and abs(abs(acc[2]) - 2.0 / 3.0) <= eps
    dif = _scatter(_abi.RT_BRDF_DIFFUSE, 0.5, 0.5, (0, 1, 0), (1, 2, 3), (0, 0, 1), 4, mode)
    assert dif.tmin == np.float32(1e-3) and dif.depth == 4 and list(dif.origin) == [1, 2, 3]


# ---- src/normal.rs:235-253 -----------------------------------------------------------------
def test_create_onb_from_z():
    pcg = O.OraclePcg()
    rf = pcg.random_floats(3 * 2000)
    for i in range(2000):
        n = np.array(rf[3 * i:3 * i + 3], dtype=np.float32)
        n = (n * (np.float32(1.0) / np.sqrt(np.float32(n.dot(n))))).astype(np.float32)
        e = (C.c_float * 9)()
        O.lib().oracle_create_onb_from_z((C.c_float * 3)(*n.tolist()), e)
        e1, e2, e3 = np.array(e[0:3]), np.array(e[3:6]), np.array(e[6:9])
        assert close(e3, n)
        assert abs(e1.dot(e2)) < EPS and abs(e2.dot(e3)) < EPS and abs(e3.dot(e1)) < EPS
        assert abs(e1.dot(e1) - 1) < EPS and abs(e2.dot(e2) - 1) < EPS and abs(e3.dot(e3) - 1) < EPS


# ---- src/transformation.rs:592-652 (T * Vector / Normal / Point) ----------------------------
def _xf(t, what, v):
    out = (C.c_float * 3)()
    O.lib().oracle_transform(O.fptr(np.ascontiguousarray(t.m)), O.fptr(np.ascontiguousarray(t.invm)), what,
                             (C.c_float * 3)(*[float(c) for c in v]), out)
    return list(out)


def test_mul_vector_normal_point():
    assert close(_xf(ra.rotation_x(np.float32(math.pi) / np.float32(3.0)), 1, (1, 1, 0)), (1, 0.5, math.sqrt(3) / 2))
    assert close(_xf(ra.scaling((2, -3, 5)), 2, (2, 1, 0)), (1, -1 / 3, 0))
    assert close(_xf(ra.translation((1, -2, 3)), 0, (-3, 2, 0)), (-2, 0, 3))
    m = ra.Transformation([[1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 8, 7], [0, 0, 0, 1]],
                          [[-3.75, 2.75, -1, 0], [5.75, -4.75, 2, 1], [-2.25, 2.25, -1, -2], [0, 0, 0, 1]])
    assert close(_xf(m, 1, (1, 2, 3)), (14, 38, 51))
    assert close(_xf(m, 0, (1, 2, 3)), (18, 46, 58))
    assert close(_xf(m, 2, (3, 2, 4)), (-8.75, 7.75, -3))


def test_mul_point_projective_divide():
    # transformation.rs:185-193: divide by w only when w != 1
    m = np.eye(4, dtype=np.float32)
    m[3, 3] = 2.0
    t = ra.Transformation(m, np.eye(4, dtype=np.float32))
    assert close(_xf(t, 0, (2, 4, 6)), (1, 2, 3))


# ---- src/camera.rs:169-202 -----------------------------------------------------------------
def _cam_rays(cam, uvs):
    c = flatten_camera(cam)
    u = np.array([p[0] for p in uvs], dtype=np.float32)
    v = np.array([p[1] for p in uvs], dtype=np.float32)
    out = (_abi.rt_ray * len(uvs))()
    O.lib().oracle_camera_fire_ray(C.byref(c), O.fptr(u), O.fptr(v), len(uvs), out)
    return list(out)


def _at(r, t):
    return [r.origin[i] + r.dir[i] * t for i in range(3)]


def test_orthogonal_camera():
    rays = _cam_rays(ra.OrthogonalCamera(2.0, ra.Transformation()), [(0, 0), (1, 0), (0, 1), (1, 1)])
    for r in rays[1:]:
        cross = np.cross(list(rays[0].dir), list(r.dir))
        assert abs(cross.dot(cross)) < EPS
    for r, p in zip(rays, [(0, 2, -1), (0, -2, -1), (0, 2, 1), (0, -2, 1)]):
        assert close(_at(r, 1.0), p)


def test_perspective_camera():
    rays = _cam_rays(ra.PerspectiveCamera(1.0, 2.0, ra.Transformation()), [(0, 0), (1, 0), (0, 1), (1, 1)])
    for r in rays[1:]:
        assert close(r.origin, rays[0].origin)
    for r, p in zip(rays, [(0, 2, -1), (0, -2, -1), (0, 2, 1), (0, -2, 1)]):
        assert close(_at(r, 1.0), p)


# ---- src/imagetracer.rs:137-183 --------------------------------------------------------------
def _fire(cam, w, h, tuples):
    c = flatten_camera(cam)
    col = np.array([t[0] for t in tuples], dtype=np.uint32)
    row = np.array([t[1] for t in tuples], dtype=np.uint32)
    up = np.array([t[2] for t in tuples], dtype=np.float32)
    vp = np.array([t[3] for t in tuples], dtype=np.float32)
    out = (_abi.rt_ray * len(tuples))()
    u32 = C.POINTER(C.c_uint32)
    O.lib().oracle_fire_ray(C.byref(c), w, h, col.ctypes.data_as(u32), row.ctypes.data_as(u32), O.fptr(up),
                            O.fptr(vp), len(tuples), out)
    return list(out)


def test_uv_sub_mapping_and_orientation():
    cam = ra.PerspectiveCamera(1.0, 2.0, ra.Transformation())
    r1, r2 = _fire(cam, 4, 2, [(0, 0, 2.5, 1.5), (2, 1, 0.5, 0.5)])
    assert close(r1.origin, r2.origin) and close(r1.dir, r2.dir)
    tl, br = _fire(cam, 4, 2, [(0, 0, 0.0, 0.0), (3, 1, 1.0, 1.0)])
    assert close(_at(tl, 1.0), (0, 2, 1)) and close(_at(br, 1.0), (0, -2, -1))


def test_image_coverage():
    cam = flatten_camera(ra.PerspectiveCamera(1.0, 2.0, ra.Transformation()))
    img, st = O.fire_all_rays(flatten_world(ra.World()), cam, ra.DummyRenderer().flatten(), 4, 2, 0, 0, 1)
    assert np.all(img == np.array([1.0, 2.0, 3.0], dtype=np.float32))
    assert st["samples"] == 8 and st["rays"] == 0


def test_oracle_modes_agree_on_demo_scene():
    """libm (what a reference binary computes on this box) vs strict-math header: the demo scene's
    only uv consumer is the plane (floor), so the two must agree except through sin/cos last-ulp
    differences; report the outlier fraction, bound the rest by the stated 1e-4 relative tolerance."""
    from rustracer_amd.scenes import demo_camera, demo_world
    w, h = 96, 64
    flat = flatten_world(demo_world())
    cam = flatten_camera(demo_camera(w, h))
    rr = ra.PathTracer(None, ra.BLACK, 10, 3, 3).flatten()
    a, sa = O.fire_all_rays(flat, cam, rr, w, h, math_mode=O.MATH_LIBM)
    b, sb = O.fire_all_rays(flat, cam, rr, w, h, math_mode=O.MATH_STRICT)
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
    bad = float(np.mean(rel.max(axis=2) > 1e-4))
    print(f"libm-vs-strict outlier pixel fraction: {bad:.5f}; rays {sa['rays']} vs {sb['rays']}")
    assert bad < 0.02


def test_shared_header_helpers_host_side_against_rusts_semantics():
    """The host bodies of rt_maxf / rt_f32_to_u32_sat (include/rt_strict_math.h) against numpy's own statement of Rust's
    f32::max (a NaN operand yields the other operand) and of the saturating `as u32` cast (NaN -> 0); the GPU suite checks
    the device bodies against the same statement (test_shared_header_helpers_against_an_independent_statement_...)."""
    rng = np.random.default_rng(11)
    n = 1 << 14
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.05, 1e-45, 3.4e38, 4294967296.0, 4294967040.0,
                         2147483648.0, 0.99999994, -1e-45, 16777216.5], dtype=np.float32)
    a = np.concatenate([np.repeat(specials, len(specials)), rng.normal(size=n).astype(np.float32)])
    b = np.concatenate([np.tile(specials, len(specials)), rng.normal(size=n).astype(np.float32)])
    got = O.math_eval(7, a, b)
    with np.errstate(all="ignore"):
        want = np.fmax(a, b)
    both_zero = (a == 0) & (b == 0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want) & ~both_zero
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    assert np.all(got[both_zero] == 0)
    x = np.concatenate([specials, rng.uniform(-10, 5e9, n).astype(np.float32), rng.uniform(0, 300, n).astype(np.float32)])
    got_u = O.math_eval(8, x).view(np.uint32)
    with np.errstate(all="ignore"):
        xd = x.astype(np.float64)
        want_u = np.where(np.isnan(xd), 0.0, np.clip(np.trunc(xd), 0.0, 4294967295.0)).astype(np.uint64).astype(np.uint32)
    assert np.array_equal(got_u, want_u)
