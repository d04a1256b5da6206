"""Randomised parity sweep: random scenes / cameras / renderers, HIP path vs the strict oracle, bit for bit.

  python tools/fuzz_parity.py FIRST_SEED N_SEEDS        (needs a GPU; prints one line per scene, exits 1 on a mismatch)

tests/test_gpu_parity.py::test_random_scenes runs a fixed handful of these seeds; this tool is for long sweeps.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rustracer_amd as ra  # noqa: E402


def random_pigment(rng, tex, bright=1.0, nasty=False):
    k = rng.integers(0, 10)
    col = tuple(float(x) * bright for x in rng.uniform(0.0, 1.0, 3))
    if nasty and rng.random() < 0.1:
        # colours no renderer should see, but f32 arithmetic defines: negative, huge, infinite, NaN
        col = tuple(float(rng.choice([-0.5, 0.0, 1e20, 3e38, float("inf"), float("nan"), 1.0, 0.3])) for _ in range(3))
        if rng.random() < 0.3:
            return ra.CheckeredPigment(col, (0.2, 0.4, 0.6), int(rng.choice([0, 1, 2**31 - 1])))
    if k < 6:
        return ra.UniformPigment(col)
    if k < 8:
        return ra.CheckeredPigment(col, tuple(float(x) * bright for x in rng.uniform(0.0, 1.0, 3)), int(rng.integers(1, 12)))
    return ra.ImagePigment(tex)


def random_transform(rng, spread, rmin, rmax, projective=False, nasty=False, axis=False):
    c = tuple(float(x) for x in rng.uniform(-spread, spread, 3))
    if axis and rng.random() < 0.3:
        c = tuple(float(round(x)) if rng.random() < 0.5 else x for x in c)      # exact zeros and integers among the translations
    r = float(rng.uniform(rmin, rmax))
    if axis and rng.random() < 0.3:
        r = -r                                                                   # mirrored shapes
    t = ra.translation(c)
    if not axis and rng.random() < 0.6:
        t = t * ra.rotation_z(float(rng.uniform(0, 360))) * ra.rotation_y(float(rng.uniform(0, 360))) * \
            ra.rotation_x(float(rng.uniform(0, 360)))
    if rng.random() < 0.5:
        t = t * ra.scaling((r, r * float(rng.uniform(0.3, 3)), r * float(rng.uniform(0.3, 3))))
    else:
        t = t * ra.scaling((r, r, r))
    if nasty and rng.random() < 0.25:
        # extreme but legal inputs: huge / tiny / flat shapes (an infinite inverse), far-away centres
        k = int(rng.integers(0, 5))
        if k == 0:
            t = t * ra.scaling((float(10 ** rng.uniform(-7, -3)),) * 3)
        elif k == 1:
            t = t * ra.scaling((float(10 ** rng.uniform(2, 7)),) * 3)
        elif k == 2:
            t = ra.translation(tuple(float(x) for x in rng.uniform(-1, 1, 3) * 10 ** rng.uniform(3, 9))) * t
        elif k == 3:
            sc = [1.0, 1.0, 1.0]
            sc[int(rng.integers(0, 3))] = 0.0
            with np.errstate(all="ignore"):
                t = t * ra.scaling(tuple(sc))
        else:
            t = t * ra.scaling((float(10 ** rng.uniform(-3, 3)), float(10 ** rng.uniform(-3, 3)), float(10 ** rng.uniform(-3, 3))))
    if projective:
        # a last row other than [0,0,0,1]: the w divide of transformation.rs:185-193 (and an unbounded shape
        # for the hierarchy)
        m = np.array(t.m, dtype=np.float64).reshape(4, 4).copy()
        m[3, 3] = float(rng.uniform(0.5, 2.0))
        m[3, int(rng.integers(0, 3))] = float(rng.uniform(-0.2, 0.2))
        if np.all(np.isfinite(m)) and abs(np.linalg.det(m)) > 1e-30:
            with np.errstate(all="ignore"):
                return ra.Transformation(m.astype(np.float32), np.linalg.inv(m).astype(np.float32))
    return t


def random_case(seed):
    rng = np.random.default_rng(seed)
    nasty = seed >= 1_000_000           # seeds from a million on also draw extreme transforms and backgrounds,
    nasty2 = seed >= 2_000_000          # from two million on extreme colours and checker steps as well
    axis = seed >= 3_000_000            # from three million on: scalings and translations only (the sparse closest-hit form
                                        # and its guard), cameras without rotation half of the time (zero direction components)
    tex = ra.HdrImage(int(rng.integers(1, 9)), int(rng.integers(1, 9)))
    tex.pixels = rng.uniform(0, 1, (tex.height, tex.width, 3)).astype(np.float32)
    size_class = int(rng.integers(0, 4))
    n_shapes = [int(rng.integers(1, 9)), int(rng.integers(9, 97)), int(rng.integers(97, 140)), int(rng.integers(140, 400))][size_class]
    spread = [3.0, 6.0, 8.0, 12.0][size_class]
    world = ra.World()
    if rng.random() < 0.7:      # an emissive enclosure, so that paths return light
        world.add(ra.Sphere(ra.scaling((60, 60, 60)), ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), random_pigment(rng, tex))))
    while len(world.shapes) < n_shapes:
        emit = random_pigment(rng, tex, 0.5, nasty2) if rng.random() < 0.15 else ra.UniformPigment(ra.BLACK)
        pig = random_pigment(rng, tex, 1.0, nasty2)
        brdf = ra.SpecularBRDF(pig) if rng.random() < 0.3 else ra.DiffuseBRDF(pig)
        mat = ra.Material(brdf, emit)
        u = rng.random()
        if u < 0.08:
            world.add(ra.Plane(random_transform(rng, spread, 0.5, 2.0, nasty=nasty, axis=axis), mat))
        elif u < 0.11 and not axis:
            world.add(ra.Sphere(random_transform(rng, spread, 0.3, 1.5, projective=True, nasty=nasty), mat))
        else:
            world.add(ra.Sphere(random_transform(rng, spread, 0.2, 1.8, nasty=nasty, axis=axis), mat))
        if rng.random() < 0.03 and len(world.shapes) < n_shapes:
            world.add(world.shapes[-1])          # exact duplicate: the earlier index wins (world.rs:38)
    w, h = int(rng.integers(1, 97)), int(rng.integers(1, 65))
    cam_t = ra.translation(tuple(float(x) for x in rng.uniform(-4, 4, 3))) * ra.rotation_z(float(rng.uniform(0, 360))) * \
        ra.rotation_y(float(rng.uniform(-40, 40)))
    if axis and rng.random() < 0.5:
        cam_t = ra.translation(tuple(float(x) for x in rng.uniform(-4, 4, 3)))
    aspect = w / h if rng.random() < 0.7 else float(rng.uniform(0.5, 2.5))
    if rng.random() < 0.75:
        cam = ra.PerspectiveCamera(float(rng.uniform(0.4, 3.0)), aspect, cam_t)
    else:
        cam = ra.OrthogonalCamera(aspect, cam_t * ra.scaling((1.0, float(rng.uniform(1, 6)), float(rng.uniform(1, 6)))))
    bg = tuple(float(x) for x in rng.uniform(0, 0.3, 3)) if rng.random() < 0.5 else ra.BLACK
    if nasty and rng.random() < 0.2:
        bg = (float("inf"), -1.0, float(rng.uniform(0, 1e30)))
    k = int(rng.integers(0, 10))
    if k == 0:
        renderer = ra.OnOffRenderer(world)
    elif k <= 2:
        renderer = ra.FlatRenderer(world, bg)
    else:
        n = int(rng.choice([1, 2, 3, 4, 5, 10]))
        d = int(rng.integers(0, 5))
        while n ** d > 1500:
            d -= 1
        renderer = ra.PathTracer(world, bg, n, d, int(rng.integers(0, 6)))
    aa = int(rng.choice([1, 1, 2, 3]))
    # keep the oracle's linear scan to seconds: an upper bound of its shape tests
    chain = 1
    if isinstance(renderer, ra.PathTracer):
        chain = sum(renderer.num_of_rays ** k for k in range(renderer.max_depth + 1))
    while w * h * aa * aa * chain * len(world.shapes) > 4e8 and (w > 8 or h > 8):
        w, h = max(w // 2, 1), max(h // 2, 1)
    return world, cam, renderer, w, h, dict(init_state=int(rng.integers(0, 2**63)), init_seq=int(rng.integers(0, 2**63)), aa=aa)


def describe(world, cam, renderer, w, h, kw):
    r = type(renderer).__name__
    if isinstance(renderer, ra.PathTracer):
        r += f"(n={renderer.num_of_rays},d={renderer.max_depth},rr={renderer.russian_roulette_limit})"
    return f"{len(world.shapes):3d} shapes {type(cam).__name__[:5]} {w}x{h} aa{kw['aa']} {r}"


def check_case(seed, case):
    """check_image (default schedule + one pass, against the oracle), then the same frame under a random pass
    schedule and team-size cap (the wavefront machinery must never change a pixel or a counter)."""
    from test_gpu_parity import assert_bit_equal, check_image, gpu_render
    world, cam, renderer, w, h = case[:5]
    g, st = check_image(world, cam, renderer, w, h, **case[5])
    if isinstance(renderer, ra.PathTracer):
        rng = np.random.default_rng(seed ^ 0x5EED)
        schedule = [(1,), (2, 3), (8,), (5, 7, 11, 13), (48, 64, 96), (200,), (1, 1, 1, 1, 1, 1)][int(rng.integers(0, 7))]
        team = ["1", "2", "4", "16", "64"][int(rng.integers(0, 5))]
        old = os.environ.get("RUSTRACER_MAX_TEAM")
        os.environ["RUSTRACER_MAX_TEAM"] = team
        try:
            g2, st2 = gpu_render(world, cam, renderer, w, h, schedule=schedule, **case[5])
        finally:
            if old is None:
                del os.environ["RUSTRACER_MAX_TEAM"]
            else:
                os.environ["RUSTRACER_MAX_TEAM"] = old
        assert_bit_equal(g2.cpu().numpy(), g, f"schedule {schedule} max_team {team}")
        for k in ("rays", "samples", "diffuse_scatters", "specular_scatters"):
            assert st2[k] == st[k], (schedule, team, k, st2[k], st[k])
    return g, st


def random_rays(rng, n, max_depth):
    from rustracer_amd.world import to_rt_ray
    rays = []
    for _ in range(n):
        o = rng.uniform(-8, 8, 3)
        u = rng.random()
        if u < 0.15:
            o = o * float(10 ** rng.uniform(1, 6))         # far origins: the discriminant is mostly rounding
        d = rng.normal(size=3)
        if u > 0.9:
            d = np.zeros(3)
            d[int(rng.integers(0, 3))] = float(rng.choice([-1, 1]))
        elif u > 0.88:
            d = np.zeros(3)
        kw = {}
        if rng.random() < 0.3:
            kw["tmin"] = float(10 ** rng.uniform(-5, 0))
        if rng.random() < 0.2:
            kw["tmax"] = float(10 ** rng.uniform(-1, 2))
        kw["depth"] = int(rng.integers(0, max_depth + 2))
        rays.append(to_rt_ray(ra.Ray(origin=tuple(o), dir=tuple(d), **kw)))
    return rays


def check_batch(seed, case):
    """World::ray_intersection and Renderer::solve on a batch of random rays (far origins, axis-parallel and zero
    directions, tmin / tmax / depth variations) against the oracle."""
    import ctypes as C
    import oracle_lib as O
    from rustracer_amd import _abi
    from rustracer_amd.world import flatten_world
    from test_gpu_parity import _hits_equal, assert_bit_equal
    world, cam, renderer = case[:3]
    rng = np.random.default_rng(seed ^ 0xBA7C4)
    md = renderer.max_depth if isinstance(renderer, ra.PathTracer) else 0
    n = 256 if len(world.shapes) > 96 else 1024
    rays = random_rays(rng, n, md)
    flat = flatten_world(world)
    g = world.ray_intersections(rays)
    o = O.world_ray_intersection(flat, rays, O.MATH_STRICT)
    for k, (a, b) in enumerate(zip(g, o)):
        try:
            _hits_equal(a, b)
        except AssertionError as e:
            raise AssertionError(f"ray {k} ({list(rays[k].origin)}, {list(rays[k].dir)}): gpu {a.hit, a.shape, a.t} oracle {b.hit, b.shape, b.t} {e}") from e
    pcgs = [(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**62)) * 2 + 1) for _ in rays]
    r = renderer.flatten()
    arr = (_abi.rt_ray * len(rays))(*rays)
    pa = (_abi.rt_pcg * len(rays))(*[_abi.rt_pcg(s, i) for s, i in pcgs])
    out = np.zeros((len(rays), 3), dtype=np.float32)
    _abi.check(_abi.lib().rt_renderer_solve(world.device().handle, C.byref(r), arr, pa, len(rays), O.fptr(out)))
    ref, new = O.renderer_solve(flat, r, rays, pcgs)
    assert_bit_equal(out, ref, "solve batch")
    assert [(pa[i].state, pa[i].inc) for i in range(len(rays))] == new, "solve batch: PCG states"


def check_frames(seed, case):
    """rt_render_frames: several frames (different cameras) in one launch, a random row-tile partition and pass
    schedule; every frame's part must equal the oracle's rows of that frame."""
    import ctypes as C
    import torch
    from rustracer_amd import _abi
    from rustracer_amd.world import flatten_camera
    from test_gpu_parity import assert_bit_equal, oracle_image
    world, cam, renderer, w, h, kw = case
    rng = np.random.default_rng(seed ^ 0xF4A3E5)
    n_frames = int(rng.choice([1, 2, 3, 4, 9, 12]))
    if n_frames > 4 and w * h > 600:
        n_frames = 3
    cams = [cam]
    for _ in range(n_frames - 1):
        t = cam.transformation * ra.rotation_z(float(rng.uniform(-0.5, 0.5))) * ra.translation(tuple(float(x) for x in rng.uniform(-0.3, 0.3, 3)))
        cams.append(type(cam)(**{**cam.__dict__, "transformation": t}))
    n_parts = int(rng.choice([1, 2, 3, 5, 8]))
    tile_rows = int(rng.choice([1, 2, 3, 7, 100]))
    part = _abi.rt_partition(int(rng.integers(0, n_parts)), n_parts, tile_rows) if n_parts > 1 else None
    schedule = [(), (0,), (3, 5)][int(rng.integers(0, 3))]
    lib, dev = _abi.lib(), world.device()
    r = renderer.flatten()
    arr = (_abi.rt_camera * n_frames)(*[flatten_camera(c) for c in cams])
    n_local = lib.rt_partition_pixels(w, h, C.byref(part) if part else None)
    outs = [torch.full((max(n_local, 1), 3), -7.0, dtype=torch.float32, device="cuda") for _ in cams]
    ptrs = (C.c_void_p * n_frames)(*[o.data_ptr() for o in outs])
    st = _abi.rt_stats()
    dev.set_schedule(schedule)
    try:
        _abi.check(lib.rt_render_frames(dev.handle, arr, n_frames, C.byref(r), w, h, kw["init_state"], kw["init_seq"], kw["aa"],
                                        C.byref(part) if part else None, ptrs, torch.cuda.current_stream().cuda_stream, C.byref(st)))
    finally:
        dev.set_schedule(())
    rows = list(range(h))
    if part is not None:
        rows = [y for y in range(h) if (y // tile_rows) % n_parts == part.part]
    assert n_local == len(rows) * w
    total_rays = 0
    for f, c in enumerate(cams):
        ref, rs = oracle_image(world, c, renderer, w, h, **kw)
        total_rays += rs["rays"]
        got = outs[f].cpu().numpy()[:n_local].reshape(len(rows), w, 3)
        assert_bit_equal(got, ref[rows], f"frame {f} of {n_frames}, part {None if part is None else (part.part, n_parts, tile_rows)}, schedule {schedule}")
    if part is None:
        assert st.rays == total_rays, (st.rays, total_rays)


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(first, first + n):
        case = random_case(seed)
        t0 = time.perf_counter()
        print(f"seed {seed}: ..   {describe(*case)}", flush=True)      # named before it runs: a fault on the card ends the process
        try:
            _, st = check_case(seed, case)
            check_batch(seed, case)
            check_frames(seed, case)
            print(f"seed {seed}: ok   {describe(*case)} rays {st['rays']} ({time.perf_counter() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: FAIL {describe(*case)}: {e}", flush=True)
    print(f"{n - bad} of {n} scenes bit-identical", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
