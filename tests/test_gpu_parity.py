"""GPU parity: the HIP render core (through the C-ABI) against the CPU oracle.

Bar: bit-exact.  The oracle runs with math_mode = STRICT (include/rt_strict_math.h, the same
definition the kernels use), so every f32 of every pixel, hit record and PCG state must be equal,
and the ray counters must match exactly.  Every test here needs a real MI355X.
"""
import ctypes as C
import os
import math

import numpy as np
import pytest

import oracle_lib as O
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera, flatten_world, to_rt_ray

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape
    neq = bits(a) != bits(b)
    # NaN payloads may differ between x86 and gfx950; NaN == NaN counts as equal
    neq &= ~(np.isnan(a) & np.isnan(b))
    assert not neq.any(), f"{what}: {int(neq.sum())} of {neq.size} floats differ, first at {np.argwhere(neq)[0]}"


def gpu_image(world, cam, renderer, w, h, init_state=42, init_seq=54, aa=1):
    img = ra.HdrImage(w, h)
    tracer = ra.ImageTracer(img, cam)
    tracer.fire_all_rays(renderer, init_state, init_seq, aa)
    return img.pixels, tracer.last_stats


def oracle_image(world, cam, renderer, w, h, init_state=42, init_seq=54, aa=1):
    return O.fire_all_rays(flatten_world(world), flatten_camera(cam), renderer.flatten(), w, h, init_state,
                           init_seq, aa, math_mode=O.MATH_STRICT)


def check_image(world, cam, renderer, w, h, **kw):
    g, gs = gpu_image(world, cam, renderer, w, h, **kw)
    o, os_ = oracle_image(world, cam, renderer, w, h, **kw)
    assert_bit_equal(g, o, "image")
    for k in ("rays", "samples", "sphere_tests", "plane_tests", "diffuse_scatters", "specular_scatters"):
        assert gs[k] == os_[k], (k, gs[k], os_[k])
    if isinstance(renderer, ra.PathTracer):
        # the same frame through the throughput schedule (one pass, one lane per pixel: the k_pass FAST variant
        # where the scene allows it; small frames would otherwise run teams)
        old = os.environ.get("RUSTRACER_MAX_TEAM")
        os.environ["RUSTRACER_MAX_TEAM"] = "1"
        try:
            g1, gs1 = gpu_render(world, cam, renderer, w, h, schedule=(0,), **kw)
        finally:
            if old is None:
                del os.environ["RUSTRACER_MAX_TEAM"]
            else:
                os.environ["RUSTRACER_MAX_TEAM"] = old
        assert_bit_equal(g1.cpu().numpy(), o, "image, one-pass schedule")
        for k in ("rays", "samples", "diffuse_scatters", "specular_scatters"):
            assert gs1[k] == os_[k], ("one-pass", k, gs1[k], os_[k])
    return g, gs


# ---- PCG (src/random.rs:56-81) ---------------------------------------------------------------
def test_pcg_known_answers_and_random_seeds():
    rng = np.random.default_rng(1)
    s = np.concatenate([[42, 38, 0, 2**64 - 1], rng.integers(0, 2**63, 1000, dtype=np.uint64)]).astype(np.uint64)
    q = np.concatenate([[54, 62, 0, 2**64 - 1], rng.integers(0, 2**63, 1000, dtype=np.uint64)]).astype(np.uint64)
    n, k = len(s), 37
    got = np.zeros(n * k, dtype=np.uint32)
    ref = np.zeros(n * k, dtype=np.uint32)
    u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    _abi.check(_abi.lib().rt_pcg_random(s.ctypes.data_as(u64), q.ctypes.data_as(u64), n, k, got.ctypes.data_as(u32)))
    O.lib().oracle_pcg_random(s.ctypes.data_as(u64), q.ctypes.data_as(u64), n, k, ref.ctypes.data_as(u32))
    assert got[:6].tolist() == [2707161783, 2068313097, 3122475824, 2211639955, 3215226955, 3421331566]
    assert np.array_equal(got, ref)


# ---- shared strict math: host == device, bit for bit -------------------------------------------
def _gpu_math(fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    _abi.check(_abi.lib().rt_strict_math_eval(fn, O.fptr(a), O.fptr(b), a.size, O.fptr(out)))
    return out


def test_strict_math_host_device_bit_equal():
    rng = np.random.default_rng(7)
    n = 1 << 20
    rf = (rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.float32) * np.float32(2.0**-32)).astype(np.float32)
    phi = (np.float32(6.2831855) * rf).astype(np.float32)       # the hot path's argument range
    wide = rng.uniform(-1000, 1000, n).astype(np.float32)
    for fn, a in ((0, phi), (1, phi), (0, wide), (1, wide)):
        assert_bit_equal(_gpu_math(fn, a), O.math_eval(fn, a), f"sin/cos fn={fn}")
    y = rng.normal(size=n).astype(np.float32)
    x = rng.normal(size=n).astype(np.float32)
    y[:8] = [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, np.nan]
    x[:8] = [1.0, 1.0, -1.0, -1.0, 0.0, -0.0, np.inf, 1.0]
    assert_bit_equal(_gpu_math(2, y, x), O.math_eval(2, y, x), "atan2")
    z = rng.uniform(-1.0000001, 1.0000001, n).astype(np.float32)
    z[:4] = [1.0, -1.0, 0.0, 1.0000001]
    assert_bit_equal(_gpu_math(3, z), O.math_eval(3, z), "acos")
    # IEEE f32 divide and sqrt incl. subnormals
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    b = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        assert_bit_equal(_gpu_math(4, a, b), O.math_eval(4, a, b), "div")
        assert_bit_equal(_gpu_math(5, np.abs(a)), O.math_eval(5, np.abs(a)), "sqrt")
        # the short form the scatter code uses (no denormal-range scaling): exact for 0, +inf and x >= 2^-96
        nrm = np.abs(a)
        nrm = np.where(np.isfinite(nrm) & (nrm >= np.float32(2.0**-96)), nrm, np.float32(0.0)).astype(np.float32)
        nrm[:6] = [0.0, 1.0, np.inf, 2.0**-96, 2.0**-32, np.float32(1.0) - np.float32(2.0**-24)]
        assert_bit_equal(_gpu_math(6, nrm), O.math_eval(5, nrm), "sqrt, normal range")
        assert_bit_equal(_gpu_math(6, rf), O.math_eval(5, rf), "sqrt(rf)")
        assert_bit_equal(_gpu_math(6, (np.float32(1.0) - rf).astype(np.float32)), O.math_eval(5, (np.float32(1.0) - rf).astype(np.float32)), "sqrt(1 - rf)")


def test_pigment_cell_estimates_stay_within_their_bounds():
    """A pigment reads only the CELL (u, v) falls in (src/material.rs:54-70, :86-96), so the kernels first look for it with
    single-precision estimates of atan2 / acos and evaluate the reference functions only where a cell boundary lies within
    the estimates' error bound (csrc/rt_device.h: hit_pigments).  That is exact as long as the bounds hold: here the
    estimates are compared with the strict functions on 2^22 points each (random directions at every scale, axis-aligned
    and near-axis points, the whole range of z up to the poles' cut-off)."""
    rng = np.random.default_rng(5)
    n = 1 << 22
    y = rng.normal(size=n).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 7, n).astype(np.float32)
    x = rng.normal(size=n).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 7, n).astype(np.float32)
    k = n // 8
    y[:k] *= np.float32(1e-6)                                   # near the x axis (both signs of x: the wrap at u = 0 / 1)
    x[k:2 * k] *= np.float32(1e-6)                              # near the y axis
    y[2 * k:3 * k] = x[2 * k:3 * k] * (1 + rng.normal(size=k).astype(np.float32) * np.float32(1e-6))   # near the octant boundary
    est = _gpu_math(9, y, x)
    ref = _gpu_math(2, y, x)
    d_atan = float(np.abs(est.astype(np.float64) - ref.astype(np.float64)).max())
    z = np.concatenate([rng.uniform(-0.999999, 0.999999, n - 4096), np.linspace(-0.999999, 0.999999, 4096)]).astype(np.float32)
    z = z[np.abs(z) < np.float32(0.999999)]
    d_acos = float(np.abs(_gpu_math(10, z).astype(np.float64) - _gpu_math(3, z).astype(np.float64)).max())
    # u and v as the kernels form them, against the reference's f32 values (shape.rs:101-108)
    u_est = est * np.float32(0.15915494309189535)
    u_est = np.where(u_est < 0, u_est + np.float32(1.0), u_est).astype(np.float32)
    u_ref = (ref / np.float32(6.2831854820251465)).astype(np.float32)
    u_ref = np.where(u_ref < 0, u_ref + np.float32(1.0), u_ref).astype(np.float32)
    away = np.abs(u_est) > 2 * 6.0e-7                           # the kernels leave the wrap at 0 to the reference functions
    d_u = float(np.abs(u_est.astype(np.float64) - u_ref.astype(np.float64))[away].max())
    v_est = _gpu_math(10, z) * np.float32(0.3183098861837907)
    v_ref = (_gpu_math(3, z) / np.float32(3.1415927410125732)).astype(np.float32)
    d_v = float(np.abs(v_est.astype(np.float64) - v_ref.astype(np.float64)).max())
    print(f"estimate errors: atan2 {d_atan:.3e} (bound 1.5e-6), acos {d_acos:.3e}, u {d_u:.3e} (bound 6e-7), v {d_v:.3e} (bound 8e-7)")
    assert d_atan <= 1.0e-6 and d_acos <= 1.5e-6 and d_u <= 4.5e-7 and d_v <= 6.0e-7


def test_shared_header_helpers_against_an_independent_statement_of_rusts_semantics():
    """rt_maxf and rt_f32_to_u32_sat live in the header the oracle shares with the kernels, so a bit-exact image test
    cannot see a bug in them; here the device versions are checked against numpy's own statement of Rust's f32::max
    (a NaN operand yields the other one) and of the saturating `as u32` cast (NaN -> 0)."""
    rng = np.random.default_rng(11)
    n = 1 << 16
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.05, 1e-45, 3.4e38, 4294967296.0, 4294967040.0,
                         2147483648.0, 0.99999994, -1e-45, 16777216.5], dtype=np.float32)
    a = np.concatenate([np.repeat(specials, len(specials)), rng.normal(size=n).astype(np.float32)])
    b = np.concatenate([np.tile(specials, len(specials)), rng.normal(size=n).astype(np.float32)])
    got = _gpu_math(7, a, b)
    with np.errstate(all="ignore"):
        want = np.fmax(a, b)                    # IEEE maxNum: the non-NaN operand; NaN only if both are
    both_zero = (a == 0) & (b == 0)             # max(+0, -0): either zero is a correct answer (and the renderers never see its sign)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want) & ~both_zero
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    assert np.all(got[both_zero] == 0)
    x = np.concatenate([specials, rng.uniform(-10, 5e9, n).astype(np.float32), rng.uniform(0, 300, n).astype(np.float32)])
    got_u = _gpu_math(8, x).view(np.uint32)
    with np.errstate(all="ignore"):
        xd = x.astype(np.float64)
        want_u = np.where(np.isnan(xd), 0.0, np.clip(np.trunc(xd), 0.0, 4294967295.0)).astype(np.uint64).astype(np.uint32)
    assert np.array_equal(got_u, want_u)


# ---- World::ray_intersection (src/shape.rs:223-568, src/world.rs:62-92) ------------------------
def _rays(rng, n):
    rays = []
    for _ in range(n):
        o = rng.uniform(-4, 4, 3)
        d = rng.normal(size=3)
        rays.append(to_rt_ray(ra.Ray(origin=tuple(o), dir=tuple(d))))
    return rays


def _hits_equal(g, o):
    assert g.hit == o.hit
    if g.hit:
        assert g.shape == o.shape
        assert_bit_equal(list(g.world_point) + list(g.normal) + list(g.uv) + [g.t],
                         list(o.world_point) + list(o.normal) + list(o.uv) + [o.t], "hit record")


def test_reference_shape_known_answers_on_gpu():
    w = ra.World()
    w.add(ra.Sphere())
    h = w.ray_intersection(ra.Ray(origin=(0, 0, 2), dir=(0, 0, -1)))
    assert h and abs(h.t - 1) < 1e-4 and list(h.world_point) == [0, 0, 1] and list(h.normal) == [0, 0, 1]
    assert w.ray_intersection(ra.Ray(origin=(0, 10, 2), dir=(0, 0, -1))) is None
    h = w.ray_intersection(ra.Ray(origin=(0, 0, 0), dir=(1, 0, 0)))   # from inside
    assert h and list(h.normal) == [-1, 0, 0] and abs(h.uv[1] - 0.5) < 1e-4
    p = ra.World()
    p.add(ra.Plane())
    h = p.ray_intersection(ra.Ray(origin=(-4.25, -7.75, 1), dir=(0, 0, -1)))
    assert h and abs(h.uv[0] - 0.75) < 1e-4 and abs(h.uv[1] - 0.25) < 1e-4
    assert p.ray_intersection(ra.Ray(origin=(0, 0, 1), dir=(1, 0, 0))) is None


def test_world_ray_intersection_random_rays():
    rng = np.random.default_rng(3)
    world = demo_world()
    world.add(ra.Sphere(ra.translation((0.5, -1, 0.3)) * ra.scaling((0.5, 1.5, 0.25)) * ra.rotation_y(0.3)))
    world.add(ra.Plane(ra.rotation_x(1.1) * ra.translation((0, 0, -2))))
    world.add(ra.Sphere())
    world.add(ra.Sphere())  # exact duplicate: tie must go to the earlier index
    rays = _rays(rng, 4000)
    rays.append(to_rt_ray(ra.Ray(origin=(0, 0, 5), dir=(0, 0, 0))))          # zero direction
    rays.append(to_rt_ray(ra.Ray(origin=(math.nan, 0, 5), dir=(0, 0, -1))))  # NaN origin (plane "hits" with t=NaN)
    rays.append(to_rt_ray(ra.Ray(origin=(0, 0, 5), dir=(0, 0, -1), tmin=1.0, tmax=4.95)))
    g = world.ray_intersections(rays)
    o = O.world_ray_intersection(flatten_world(world), rays, O.MATH_STRICT)
    for a, b in zip(g, o):
        _hits_equal(a, b)
    assert sum(h.hit for h in g) > 1000


@pytest.mark.parametrize("octants", ["0", "1"])
def test_hierarchy_returns_the_linear_scans_winner_on_adversarial_rays(octants, monkeypatch):
    """Scenes of more than 96 shapes are traversed through the exact-winner hierarchy (rt_device.h): every hit
    record must equal the oracle's linear scan, also where the f32 discriminant is mostly rounding error
    (far origins, tiny far spheres, grazing rays), on ties (duplicates) and for unbounded / projective shapes.
    Both with one layout of the tree and with the eight near-to-far layouts large worlds get."""
    monkeypatch.setenv("RUSTRACER_BVH_OCTANTS", octants)
    rng = np.random.default_rng(11)
    world = ra.World()
    world.add(ra.Sphere(ra.scaling((400, 400, 400))))                       # everything is inside this one
    world.add(ra.Plane(ra.translation((0, 0, -3))))
    centres, radii = [], []
    for k in range(420):
        c = rng.uniform(-60, 60, 3)
        r = float(10 ** rng.uniform(-1.3, 0.7))
        if k % 7 == 0:
            t = ra.translation(tuple(c)) * ra.rotation_z(float(rng.uniform(0, 3))) * ra.rotation_x(float(rng.uniform(0, 3))) * \
                ra.scaling((r, r * float(rng.uniform(0.2, 3)), r * float(rng.uniform(0.2, 3))))
        else:
            t = ra.translation(tuple(c)) * ra.scaling((r, r, r))
        world.add(ra.Sphere(t))
        if k % 50 == 0:
            world.add(ra.Sphere(t))                                          # exact duplicate: the earlier index wins
        centres.append(c)
        radii.append(r)
    m = np.eye(4, dtype=np.float32)
    m[3, 3] = 2.0
    m[3, 0] = 0.25
    world.add(ra.Sphere(ra.Transformation(m, np.linalg.inv(m.astype(np.float64)).astype(np.float32))))   # projective: unbounded
    world.add(ra.Plane(ra.rotation_x(1.1) * ra.translation((0, 0, -20))))
    centres, radii = np.array(centres), np.array(radii)
    rays = _rays(rng, 3000)
    def aimed(origin, k, miss):
        c, r = centres[k], radii[k]
        to_c = c - origin
        perp = np.cross(to_c, rng.normal(size=3))
        perp /= np.linalg.norm(perp)
        return to_rt_ray(ra.Ray(origin=tuple(origin), dir=tuple(c + perp * r * miss - origin)))
    for _ in range(6000):                                                    # grazing: through the limb, just in / just out
        k = int(rng.integers(0, len(radii)))
        origin = rng.uniform(-70, 70, 3) * (1.0 if rng.random() < 0.7 else float(10 ** rng.uniform(0, 3)))
        rays.append(aimed(origin, k, 1.0 + float(rng.choice([-1, 1])) * float(10 ** rng.uniform(-7, -1))))
    for _ in range(1500):                                                    # origins on a surface, as scattered rays start
        k = int(rng.integers(0, len(radii)))
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        rays.append(to_rt_ray(ra.Ray(origin=tuple(centres[k] + n * radii[k]), dir=tuple(rng.normal(size=3)), tmin=1e-3)))
    for _ in range(500):                                                     # axis-parallel directions, zeros included
        d = np.zeros(3)
        d[int(rng.integers(0, 3))] = float(rng.choice([-1, 1]))
        rays.append(to_rt_ray(ra.Ray(origin=tuple(rng.uniform(-60, 60, 3)), dir=tuple(d))))
    rays.append(to_rt_ray(ra.Ray(origin=(0, 0, 5), dir=(0, 0, 0))))
    rays.append(to_rt_ray(ra.Ray(origin=(math.nan, 0, 5), dir=(0, 0, -1))))
    rays.append(to_rt_ray(ra.Ray(origin=(0, 0, 50), dir=(0, 0, -1), tmin=1.0, tmax=30.0)))
    g = world.ray_intersections(rays)
    nn, nu = C.c_uint32(), C.c_uint32()
    _abi.check(_abi.lib().rt_scene_hierarchy_info(world.device().handle, C.byref(nn), C.byref(nu)))
    assert nn.value > 200 and nu.value == 3, (nn.value, nu.value)   # in use; outside it: 2 planes + the projective sphere
    o = O.world_ray_intersection(flatten_world(world), rays, O.MATH_STRICT)
    for a, b in zip(g, o):
        _hits_equal(a, b)
    assert len({h.shape for h in g if h.hit}) > 200
    # the same small spheres alone (nothing encloses them, no plane in front): seen from far away the f32
    # discriminant is mostly rounding error and the reference reports hits well outside the geometric sphere
    bare = ra.World()
    for s in world.shapes[2:-2]:
        bare.add(s)
    far = []
    for _ in range(6000):
        k = int(rng.integers(0, len(radii)))
        origin = centres[k] + rng.normal(size=3) * float(10 ** rng.uniform(1.5, 5))
        far.append(aimed(origin, k, 1.0 + float(rng.choice([-1, 1])) * float(10 ** rng.uniform(-6, 0.5))))
    g = bare.ray_intersections(far)
    o = O.world_ray_intersection(flatten_world(bare), far, O.MATH_STRICT)
    for a, b in zip(g, o):
        _hits_equal(a, b)
    assert 500 < sum(h.hit for h in g) < 5500


def test_non_affine_matrix_takes_divide_branch():
    m = np.eye(4, dtype=np.float32)
    m[3, 3] = 2.0
    m[3, 0] = 0.25
    inv = np.linalg.inv(m.astype(np.float64)).astype(np.float32)
    world = ra.World()
    world.add(ra.Sphere(ra.Transformation(m, inv)))
    world.add(ra.Plane(ra.translation((0, 0, -1))))
    rays = _rays(np.random.default_rng(5), 1000)
    g = world.ray_intersections(rays)
    o = O.world_ray_intersection(flatten_world(world), rays, O.MATH_STRICT)
    for a, b in zip(g, o):
        _hits_equal(a, b)


def test_empty_world_and_empty_batch():
    w = ra.World()
    assert w.ray_intersection(ra.Ray()) is None
    assert w.ray_intersections([]) == []


# ---- Renderer::solve (src/render.rs:221-313) ---------------------------------------------------
def test_furnace_on_gpu_matches_oracle_and_closed_form():
    pcg = O.OraclePcg()
    gpcg = ra.Pcg()
    for _ in range(10):
        emitted = pcg.random_float()
        reflectance = float(np.float32(pcg.random_float()) * np.float32(0.9))
        gpcg.state, gpcg.inc = pcg.state, pcg.inc
        mat = ra.Material(ra.DiffuseBRDF(ra.UniformPigment((reflectance,) * 3)), ra.UniformPigment((emitted,) * 3))
        w = ra.World()
        w.add(ra.Sphere(ra.Transformation(), mat))
        pt = ra.PathTracer(w, ra.BLACK, 1, 100, 101)     # max_depth 100: global-memory recursion stack
        got = pt.solve(ra.Ray(), gpcg)
        ref, new = O.renderer_solve(flatten_world(w), pt.flatten(), [to_rt_ray(ra.Ray())], [(pcg.state, pcg.inc)])
        pcg.state, pcg.inc = new[0]
        assert_bit_equal(got, ref[0], "furnace radiance")
        assert (gpcg.state, gpcg.inc) == (pcg.state, pcg.inc)
        assert abs(got[0] - emitted / (1.0 - reflectance)) < 1e-4


def test_flat_onoff_background_known_answers_on_gpu():
    red, green, blue = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    mat = ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment(red, blue, 2)), ra.UniformPigment(green))
    w = ra.World()
    w.add(ra.Sphere(ra.Transformation(), mat))
    flat = ra.FlatRenderer(w, ra.BLACK)
    pcg = ra.Pcg()
    assert flat.solve(ra.Ray(origin=(-2, 3, 0)), pcg) == (0, 0, 0)
    assert flat.solve(ra.Ray(origin=(-2, 0.5, 0.5)), pcg) == (1, 1, 0)
    assert flat.solve(ra.Ray(origin=(-2, -0.5, 0.5)), pcg) == (0, 1, 1)
    onoff = ra.OnOffRenderer(w, ra.BLACK, ra.WHITE)
    assert onoff.solve(ra.Ray(origin=(-2, 3, 0)), pcg) == (0, 0, 0)
    assert onoff.solve(ra.Ray(origin=(-2, 0, 0)), pcg) == (1, 1, 1)
    w2 = ra.World()
    w2.add(ra.Sphere(ra.translation((2, 0, 0))))
    assert ra.PathTracer(w2, ra.BLACK, 1000, 1000, 0).solve(ra.Ray(), pcg) == (0, 0, 0)
    assert ra.DummyRenderer().solve(ra.Ray(), pcg) == (1, 2, 3)


def test_solve_batch_random_rays_pathtracer():
    rng = np.random.default_rng(11)
    world = demo_world()
    flat = flatten_world(world)
    rays = _rays(rng, 512)
    pcgs = [(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**62)) * 2 + 1) for _ in rays]
    for n, d, rr in [(10, 3, 3), (3, 5, 2), (1, 2, 0), (0, 3, 3), (4, 0, 0), (4, 1, 5)]:
        r = ra.PathTracer(world, (0.1, 0.2, 0.3), n, d, rr).flatten()
        arr = (_abi.rt_ray * len(rays))(*rays)
        pa = (_abi.rt_pcg * len(rays))(*[_abi.rt_pcg(s, i) for s, i in pcgs])
        out = np.zeros((len(rays), 3), dtype=np.float32)
        _abi.check(_abi.lib().rt_renderer_solve(world.device().handle, C.byref(r), arr, pa, len(rays), O.fptr(out)))
        ref, new = O.renderer_solve(flat, r, rays, pcgs)
        assert_bit_equal(out, ref, f"solve n={n} d={d} rr={rr}")
        assert [(pa[i].state, pa[i].inc) for i in range(len(rays))] == new


# ---- ImageTracer::fire_all_rays: BASELINE configs at oracle-sized resolutions ------------------
def test_c1_onoff_mask_pixel_exact_640x480():
    world = demo_world()
    g, st = check_image(world, demo_camera(640, 480), ra.OnOffRenderer(world, ra.BLACK, ra.WHITE), 640, 480)
    assert st["samples"] == 640 * 480 and set(np.unique(g).tolist()) <= {0.0, 1.0}


def test_c2_flat_aa4_demo():
    world = demo_world()
    check_image(world, demo_camera(320, 240), ra.FlatRenderer(world, ra.BLACK), 320, 240, aa=4)


def test_flat_and_onoff_frames_through_the_general_kernel_too(monkeypatch):
    """k_simple renders these frames by default; k_pass must keep giving the same pixels (it serves
    rt_renderer_solve for these renderers)."""
    monkeypatch.setenv("RUSTRACER_NO_SIMPLE", "1")
    world = demo_world()
    check_image(world, demo_camera(160, 120), ra.FlatRenderer(world, ra.BLACK), 160, 120, aa=3)
    check_image(world, demo_camera(160, 120), ra.OnOffRenderer(world, ra.BLACK, ra.WHITE), 160, 120)


def test_c3_pathtracer_demo_small():
    world = demo_world()
    g, st = check_image(world, demo_camera(192, 108), ra.PathTracer(world, ra.BLACK, 10, 3, 3), 192, 108)
    assert st["rays"] > 192 * 108 * 10


@pytest.mark.parametrize("n,d,rr", [(0, 3, 3), (4, 0, 0), (1, 2, 0), (4, 1, 5), (2, 4, 1), (7, 2, 2), (2, 7, 3), (1, 12, 9)])
def test_pathtracer_parameter_corners_image(n, d, rr):
    """num_of_rays 0, max_depth 0, roulette from depth 0 / never reached, recursions too deep for the LDS stack (max_depth
    7 and 12: saved levels in HBM): through the latency schedule and (check_image) the FAST variant."""
    world = demo_world()
    check_image(world, demo_camera(96, 54), ra.PathTracer(world, (0.02, 0.03, 0.05), n, d, rr), 96, 54, init_state=9, init_seq=11)


def test_pathtracer_aa2_other_seeds_orthogonal():
    world = demo_world()
    check_image(world, demo_camera(96, 64, angle_deg=35.0, orthogonal=True),
                ra.PathTracer(world, (0.05, 0.0, 0.1), 3, 4, 2), 96, 64, init_state=7, init_seq=2**40 + 3, aa=2)


def test_image_pigment_and_checkered_sphere():
    rng = np.random.default_rng(2)
    tex = ra.HdrImage(17, 9)
    tex.pixels = rng.uniform(0, 1, (9, 17, 3)).astype(np.float32)
    world = ra.World()
    world.add(ra.Sphere(ra.scaling((50, 50, 50)), ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.ImagePigment(tex))))
    world.add(ra.Plane(ra.translation((0, 0, -1)), ra.Material(ra.DiffuseBRDF(ra.ImagePigment(tex)), ra.UniformPigment(ra.BLACK))))
    world.add(ra.Sphere(ra.Transformation(), ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), 6)), ra.UniformPigment(ra.BLACK))))
    world.add(ra.Sphere(ra.translation((0, 1.5, 0)), ra.Material(ra.SpecularBRDF(ra.CheckeredPigment((0.8, 0.8, 0.8), (0.3, 0.3, 0.9), 4)), ra.UniformPigment(ra.BLACK))))
    cam = ra.PerspectiveCamera(1.0, 1.5, ra.translation((-3, 0.5, 0.3)))
    check_image(world, cam, ra.FlatRenderer(world, ra.BLACK), 150, 100, aa=2)
    check_image(world, cam, ra.PathTracer(world, ra.BLACK, 4, 3, 2), 150, 100)


def test_dummy_renderer_coverage_and_degenerate_sizes():
    cam = ra.PerspectiveCamera(1.0, 2.0, ra.Transformation())
    g, st = gpu_image(ra.World(), cam, ra.DummyRenderer(), 4, 2, 0, 0, 1)   # imagetracer.rs:148-163
    assert np.all(g == np.array([1, 2, 3], dtype=np.float32)) and st["rays"] == 0
    g, _ = gpu_image(ra.World(), cam, ra.DummyRenderer(), 0, 0)
    assert g.size == 0
    world = demo_world()
    check_image(world, demo_camera(1, 1), ra.PathTracer(world, ra.BLACK, 2, 2, 1), 1, 1)
    check_image(world, demo_camera(67, 3), ra.PathTracer(world, ra.BLACK, 2, 2, 1), 67, 3, aa=3)


def test_row_tile_partition_and_unpack_reassemble_the_image():
    import torch
    world = demo_world()
    w, h = 160, 90
    cam, rr = flatten_camera(demo_camera(w, h)), ra.PathTracer(world, ra.BLACK, 3, 2, 1).flatten()
    whole, _ = gpu_image(world, demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 3, 2, 1), w, h)
    lib = _abi.lib()
    for n_parts, tile_rows in [(2, 1), (3, 4), (8, 2), (4, 100)]:
        sizes = [lib.rt_partition_pixels(w, h, C.byref(_abi.rt_partition(p, n_parts, tile_rows))) for p in range(n_parts)]
        assert sum(sizes) == w * h
        stride = max(sizes)
        parts = torch.zeros((n_parts, stride, 3), dtype=torch.float32, device="cuda")
        for p in range(n_parts):
            part = _abi.rt_partition(p, n_parts, tile_rows)
            _abi.check(lib.rt_render(world.device().handle, C.byref(cam), C.byref(rr), w, h, 42, 54, 1, C.byref(part),
                                     parts[p].data_ptr(), torch.cuda.current_stream().cuda_stream, None))
        image = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        _abi.check(lib.rt_unpack_parts(parts.data_ptr(), stride, n_parts, tile_rows, w, h, image.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert_bit_equal(image.cpu().numpy(), whole, f"partition {n_parts}x{tile_rows}")


def test_errors_are_reported_not_swallowed():
    lib = _abi.lib()
    world = demo_world()
    cam, rr = flatten_camera(demo_camera(8, 8)), ra.PathTracer(world, ra.BLACK, 1, 1, 1).flatten()
    assert lib.rt_render(world.device().handle, C.byref(cam), C.byref(rr), 8, 8, 42, 54, 0, None, None, None, None) != 0
    assert b"antialiasing" in lib.rt_last_error()
    bad = flatten_world(world)
    bad.shapes[0].material = 99
    assert not lib.rt_scene_create(*bad.args())
    assert b"material index" in lib.rt_last_error()


# ---- pass schedules, team sizes, large scenes (the wavefront machinery must never change a pixel) ----
def gpu_render(world, cam, renderer, w, h, init_state=42, init_seq=54, aa=1, schedule=(), stream=None, out=None):
    """rt_render with the scene resident in HBM (the path bench.py times)."""
    import torch
    dev = world.device()
    dev.set_schedule(schedule)
    c, r = flatten_camera(cam), renderer.flatten()
    if out is None:
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    st = _abi.rt_stats()
    s = (stream or torch.cuda.current_stream()).cuda_stream
    _abi.check(_abi.lib().rt_render(dev.handle, C.byref(c), C.byref(r), w, h, init_state, init_seq, aa, None,
                                    out.data_ptr(), s, C.byref(st) if stream is None else None))
    dev.set_schedule(())
    return out, st.as_dict()


COUNTERS = ("rays", "samples", "sphere_tests", "plane_tests", "diffuse_scatters", "specular_scatters")


@pytest.mark.parametrize("size", [(24, 16), (200, 112), (512, 288)])
def test_pass_schedules_and_team_sizes_never_change_a_pixel(size, monkeypatch):
    w, h = size
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, rs = oracle_image(world, cam, rr, w, h)
    for schedule in [(), (0,), (1,), (3, 5, 7), (40,), (200, 400)]:
        for max_team in (None, "1", "2", "8"):
            if max_team is None:
                monkeypatch.delenv("RUSTRACER_MAX_TEAM", raising=False)
            else:
                monkeypatch.setenv("RUSTRACER_MAX_TEAM", max_team)
            g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
            assert_bit_equal(g.cpu().numpy(), ref, f"{size} schedule={schedule} max_team={max_team}")
            for k in COUNTERS:
                assert gs[k] == rs[k], (size, schedule, max_team, k, gs[k], rs[k])


def test_antialiased_pixels_parked_between_samples():
    world = demo_world()
    w, h = 64, 36
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 3, 2, 1)
    ref, rs = oracle_image(world, cam, rr, w, h, aa=3)
    for schedule in [(), (1,), (2, 3), (0,)]:
        g, gs = gpu_render(world, cam, rr, w, h, aa=3, schedule=schedule)
        assert_bit_equal(g.cpu().numpy(), ref, f"aa=3 schedule={schedule}")
        assert gs["rays"] == rs["rays"] and gs["samples"] == rs["samples"]


def test_solve_batch_with_short_passes():
    rng = np.random.default_rng(5)
    world = demo_world()
    flat = flatten_world(world)
    rays = _rays(rng, 300)
    pcgs = [(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**62)) * 2 + 1) for _ in rays]
    r = ra.PathTracer(world, (0.1, 0.2, 0.3), 6, 3, 2).flatten()
    ref, new = O.renderer_solve(flat, r, rays, pcgs)
    for schedule in [(2,), (5, 9)]:
        world.device().set_schedule(schedule)
        arr = (_abi.rt_ray * len(rays))(*rays)
        pa = (_abi.rt_pcg * len(rays))(*[_abi.rt_pcg(s, i) for s, i in pcgs])
        out = np.zeros((len(rays), 3), dtype=np.float32)
        _abi.check(_abi.lib().rt_renderer_solve(world.device().handle, C.byref(r), arr, pa, len(rays), O.fptr(out)))
        world.device().set_schedule(())
        assert_bit_equal(out, ref, f"solve schedule={schedule}")
        assert [(pa[i].state, pa[i].inc) for i in range(len(rays))] == new


def test_c3_headline_frame_1920x1080_bit_exact():
    """The whole frame bench.py times (BASELINE config 3): every f32 of every pixel and every counter."""
    w, h = 1920, 1080
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, rs = oracle_image(world, cam, rr, w, h)
    assert rs["rays"] == 44180520
    for schedule in [(), (0,)]:
        g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
        assert_bit_equal(g.cpu().numpy(), ref, f"headline frame schedule={schedule}")
        for k in COUNTERS:
            assert gs[k] == rs[k], (schedule, k)


def test_frames_in_flight_on_separate_streams():
    import torch
    w, h = 320, 180
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, _ = oracle_image(world, cam, rr, w, h)
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in range(9)]
    for i, o in enumerate(outs):       # 9 frames over 3 streams: every work slot is reused
        gpu_render(world, cam, rr, w, h, schedule=(0,) if i % 2 else (), stream=streams[i % 3], out=o)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert_bit_equal(o.cpu().numpy(), ref, f"frame {i} in flight")


def test_c4_flake_228_shapes_scene_in_hbm():
    """BASELINE config 4 at an oracle-sized resolution: 228 shapes (scan table read from HBM with
    scalar loads, not LDS), mirror spheres; plain and with checkered + image pigments."""
    from rustracer_amd.scenes import flake_camera, flake_world
    w, h = 96, 54
    for textured in (False, True):
        world = flake_world(textured)
        assert len(world.shapes) == 228
        cam = flake_camera(w, h)
        check_image(world, cam, ra.FlatRenderer(world, ra.BLACK), w, h, aa=2)
        rr = ra.PathTracer(world, ra.BLACK, 4, 3, 2)
        ref, rs = oracle_image(world, cam, rr, w, h)
        for schedule in [(), (0,), (6, 11)]:
            g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
            assert_bit_equal(g.cpu().numpy(), ref, f"flake textured={textured} schedule={schedule}")
            for k in COUNTERS:
                assert gs[k] == rs[k], (textured, schedule, k)


def test_c5_random_spheres_10002_shapes():
    """BASELINE config 5's scene at an oracle-sized resolution."""
    from rustracer_amd.scenes import random_spheres_camera, random_spheres_world
    world = random_spheres_world(10000)
    assert len(world.shapes) == 10002
    w, h = 64, 36
    cam = random_spheres_camera(w, h)
    check_image(world, cam, ra.OnOffRenderer(world, ra.BLACK, ra.WHITE), w, h)
    rr = ra.PathTracer(world, ra.BLACK, 3, 2, 2)
    ref, rs = oracle_image(world, cam, rr, w, h)
    g, gs = gpu_render(world, cam, rr, w, h)
    assert_bit_equal(g.cpu().numpy(), ref, "random spheres")
    for k in COUNTERS:
        assert gs[k] == rs[k], k


def test_hierarchy_scene_of_ellipsoids_mirrors_and_pigments_path_traced():
    """A 300-shape scene (rotated / non-uniformly scaled spheres, mirrors, checkered and image pigments, two
    planes, an enclosing sky) through every image kernel with the hierarchy on: flat (k_simple), path tracer in
    the latency schedule and in one pass."""
    rng = np.random.default_rng(23)
    tex = ra.HdrImage(8, 5)
    tex.pixels = rng.uniform(0, 1, (5, 8, 3)).astype(np.float32)
    world = ra.World()
    world.add(ra.Sphere(ra.scaling((80, 80, 80)), ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.UniformPigment((1.0, 0.9, 0.7)))))
    world.add(ra.Plane(ra.translation((0, 0, -1)), ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment((0.3, 0.5, 0.1), (0.1, 0.2, 0.5), 4)), ra.UniformPigment(ra.BLACK))))
    for k in range(297):
        c = (float(rng.uniform(-6, 14)), float(rng.uniform(-10, 10)), float(rng.uniform(-0.8, 4)))
        r = float(rng.uniform(0.15, 0.9))
        t = ra.translation(c) * ra.rotation_z(float(rng.uniform(0, 3))) * ra.rotation_y(float(rng.uniform(0, 3))) * \
            ra.scaling((r, r * float(rng.uniform(0.4, 2.0)), r * float(rng.uniform(0.4, 2.0))))
        col = tuple(float(x) for x in rng.uniform(0.1, 0.9, 3))
        if k % 5 == 0:
            brdf = ra.SpecularBRDF(ra.UniformPigment(col))
        elif k % 5 == 1:
            brdf = ra.DiffuseBRDF(ra.CheckeredPigment(col, (0.9, 0.9, 0.9), 6))
        elif k % 5 == 2:
            brdf = ra.DiffuseBRDF(ra.ImagePigment(tex))
        else:
            brdf = ra.DiffuseBRDF(ra.UniformPigment(col))
        world.add(ra.Sphere(t, ra.Material(brdf, ra.UniformPigment(ra.BLACK))))
    world.add(ra.Plane(ra.rotation_y(1.3) * ra.translation((0, 0, -30))))
    assert len(world.shapes) == 300
    cam = ra.PerspectiveCamera(1.0, 1.6, ra.translation((-4.0, 0.0, 1.5)))
    check_image(world, cam, ra.FlatRenderer(world, ra.BLACK), 160, 100, aa=2)
    check_image(world, cam, ra.PathTracer(world, ra.BLACK, 4, 3, 2), 160, 100)
    check_image(world, cam, ra.PathTracer(world, ra.BLACK, 3, 2, 1), 96, 60, aa=2, init_state=5, init_seq=77)


@pytest.mark.parametrize("n,d,rr_limit", [(2, 5, 0), (3, 4, 1), (2, 4, 2), (4, 3, 0)])
def test_mirror_chains_over_a_hierarchy_with_the_roulette_at_every_level(n, d, rr_limit):
    """Mirror templates (k_pass: a hit opened by a repeated ray inherits what its sibling's children learned): dark
    mirrors, so that the roulette kills most first children, over a 200-sphere hierarchy -- a hit opened by a repeated
    ray whose siblings all died starts from nothing and must void what the level below it had kept.  (The first
    version of the rule voided only below hits opened by new rays and read a stale / never written entry here.)"""
    rng = np.random.default_rng(101 + n)
    world = ra.World()
    world.add(ra.Sphere(ra.scaling((40, 40, 40)), ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.UniformPigment((0.9, 1.0, 0.8)))))
    for k in range(200):
        c = (float(rng.uniform(-3, 3)), float(rng.uniform(-3.5, 3.5)), float(rng.uniform(-2.5, 2.5)))
        r = float(rng.uniform(0.5, 1.2))
        lum = float(rng.choice([0.5, 0.8, 0.95, 0.99]))
        col = (lum, lum * float(rng.uniform(0.2, 1.0)), lum * float(rng.uniform(0.2, 1.0)))
        brdf = ra.SpecularBRDF(ra.UniformPigment(col)) if k % 10 != 0 else ra.DiffuseBRDF(ra.UniformPigment(col))
        world.add(ra.Sphere(ra.translation(c) * ra.scaling((r, r, r)), ra.Material(brdf, ra.UniformPigment(ra.BLACK))))
    cam = ra.PerspectiveCamera(1.0, 1.5, ra.translation((-6.0, 0.0, 0.0)))
    w, h = 144, 96
    rr = ra.PathTracer(world, ra.BLACK, n, d, rr_limit)
    ref, rs = oracle_image(world, cam, rr, w, h, init_state=9 + d, init_seq=4 + n)
    assert rs["specular_scatters"] > w * h, rs
    for schedule in [(), (0,)]:
        g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule, init_state=9 + d, init_seq=4 + n)
        assert_bit_equal(g.cpu().numpy(), ref, f"dark mirrors n={n} d={d} rr={rr_limit} schedule={schedule}")
        for k in COUNTERS:
            assert gs[k] == rs[k], (schedule, k)


def test_random_scenes():
    """Random scenes (1-400 shapes: LDS and HBM tables, hierarchy, projective matrices, duplicates, every pigment and
    BRDF), cameras, renderers and seeds from tools/fuzz_parity.py, each also under a random pass schedule / team cap
    and through the batch World::ray_intersection / Renderer::solve calls on adversarial rays, every other one as a
    multi-frame launch over a random row-tile partition; sweeps of 11 750 ordinary and 7 200 extreme seeds ran clean on the MI355X."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    # seeds from a million on add extreme transforms (huge, tiny, flat: infinite / NaN matrices) and colours; the listed
    # ones had a plane hit at t = NaN that the hierarchy ordered differently from the linear scan
    seeds = list(range(1000, 1060)) + list(range(1000000, 1000060)) + [1000076, 1000165, 1000221, 1000347, 1000651] + \
        list(range(2000000, 2000030)) + \
        list(range(3000000, 3000040))       # ... two million on: negative / huge / infinite / NaN colours; three million on: worlds of
                                            # scalings and translations only (the sparse closest-hit form and its ray guard)
    import warnings
    for seed in seeds:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            case = fuzz_parity.random_case(seed)
        try:
            fuzz_parity.check_case(seed, case)
            fuzz_parity.check_batch(seed, case)
            if seed % 2 == 0:
                fuzz_parity.check_frames(seed, case)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} ({fuzz_parity.describe(*case)}): {e}") from e
    # worlds that go through the hierarchy once more, with the eight near-to-far layouts forced (by themselves only
    # trees beyond 16 KiB of nodes get them)
    os.environ["RUSTRACER_BVH_OCTANTS"] = "1"
    try:
        for seed in seeds:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                case = fuzz_parity.random_case(seed)
            if len(case[0].shapes) <= 96:
                continue
            try:
                fuzz_parity.check_case(seed, case)
                fuzz_parity.check_batch(seed, case)
            except AssertionError as e:
                raise AssertionError(f"seed {seed}, octant layouts ({fuzz_parity.describe(*case)}): {e}") from e
    finally:
        del os.environ["RUSTRACER_BVH_OCTANTS"]


def test_render_frames_animation_in_one_launch():
    """rt_render_frames: frames with different cameras share one launch; each equals its own fire_all_rays."""
    import torch
    w, h = 150, 90               # 13500 pixels per frame: chunks of 64 run across frame boundaries
    world = demo_world()
    rr = ra.PathTracer(world, ra.BLACK, 5, 3, 2)
    angles = [0.0, 10.0, 200.0, 90.0, 45.0, 30.0, 60.0, 75.0, 120.0, 300.0]   # > RT_INLINE_FRAMES: uploaded frame table
    all_cams = [demo_camera(w, h, a) for a in angles]
    all_refs = [oracle_image(world, c, rr, w, h) for c in all_cams]
    lib, dev = _abi.lib(), world.device()
    r = rr.flatten()
    for schedule, n_use in [((), 10), ((0,), 5), ((4, 9), 10)]:
        for part in [None, _abi.rt_partition(1, 3, 2)]:
            dev.set_schedule(schedule)
            cams, refs = all_cams[:n_use], all_refs[:n_use]
            arr = (_abi.rt_camera * len(cams))(*[flatten_camera(c) for c in cams])
            n_local = lib.rt_partition_pixels(w, h, C.byref(part) if part else None)
            outs = [torch.zeros((n_local, 3), dtype=torch.float32, device="cuda") for _ in cams]
            ptrs = (C.c_void_p * len(cams))(*[o.data_ptr() for o in outs])
            st = _abi.rt_stats()
            _abi.check(lib.rt_render_frames(dev.handle, arr, len(cams), C.byref(r), w, h, 42, 54, 1,
                                            C.byref(part) if part else None, ptrs,
                                            torch.cuda.current_stream().cuda_stream, C.byref(st)))
            dev.set_schedule(())
            rows = list(range(h)) if part is None else [y for y in range(h) if (y // 2) % 3 == 1]
            for f, (ref, rs) in enumerate(refs):
                assert_bit_equal(outs[f].cpu().numpy().reshape(len(rows), w, 3), ref[rows], f"frame {f} schedule={schedule}")
            if part is None:
                assert st.rays == sum(rs["rays"] for _, rs in refs)


def test_cli_demo_matches_the_references_own_render_statistically(tmp_path):
    """End to end through the CLI mirror: `demo --width 1920 --height 1080 --anti-aliasing 3` against the
    reference's examples/demo.png (condensed by tools/make_golden_demo_png.py).  The PNG was produced by
    an unknown earlier version with a different random stream, so noisy pixels differ individually: the
    noise-free sky rows must be identical and 40x40 block means must agree within 1/255 (measured 0.84
    max, 0.04 mean with the CPU checker)."""
    from PIL import Image
    from rustracer_amd import cli
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_png_soft.npz"))
    out = str(tmp_path / "demo.png")
    assert cli.main(["demo", out, "--width", "1920", "--height", "1080", "--anti-aliasing", "3", "--output-pfm"]) == 0
    got = np.asarray(Image.open(out).convert("RGB")).astype(np.float64)
    assert (got[:int(g["sky_rows"])] == g["sky_rgb"]).all()
    blocks = got.reshape(27, 40, 48, 40, 3).mean((1, 3))
    d = np.abs(blocks - g["block40"])
    assert d.max() < 1.0 and d.mean() < 0.08, (d.max(), d.mean())
    # --output-pfm wrote the HDR image next to it; `convert` reproduces the PNG from it byte for byte
    out2 = str(tmp_path / "again.png")
    assert cli.main(["convert", str(tmp_path / "demo.pfm"), out2]) == 0
    assert np.array_equal(np.asarray(Image.open(out2)), np.asarray(Image.open(out)))


def test_cli_render_flake_matches_the_references_own_render_statistically(tmp_path):
    """End to end through the scene-file parser, the exact-winner hierarchy (228 shapes) and tone mapping:
    `render --width 1280 --height 720 --anti-aliasing 3 flake.yml` (README.md:276; the scene file is the emitted
    counterpart of `cue eval flake.cue`) against the reference's examples/flake.png, condensed by
    tools/make_golden_flake_png.py.  Random streams of the version that made the PNG are unknown, so pixels
    differ individually: 40x40 block means must agree within 1/255 (measured 0.47 max, 0.046 mean) and the
    whole-image mean within 0.02 (measured 0.008)."""
    from PIL import Image
    from rustracer_amd import cli
    from rustracer_amd.scenes import flake_yml
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flake_png_soft.npz"))
    scene = tmp_path / "flake.yml"
    scene.write_text(flake_yml())
    out = str(tmp_path / "flake.png")
    assert cli.main(["render", str(scene), out, "--width", "1280", "--height", "720", "--anti-aliasing", "3"]) == 0
    got = np.asarray(Image.open(out).convert("RGB")).astype(np.float64)
    d = np.abs(got.reshape(18, 40, 32, 40, 3).mean((1, 3)) - g["block40"])
    assert d.max() < 1.0 and d.mean() < 0.1, (d.max(), d.mean())
    assert np.abs(got.mean((0, 1)) - g["mean_rgb"]).max() < 0.02
