"""GPU parity at BASELINE.json's own parameters (configs 2-5), the GPU-vs-libm outlier bound on the headline frame,
and the internal multi-GPU fan-out of the drop-in call.

C2 runs at its full size.  C4 / C5 use the configs' renderer parameters (pathtracer n=10, depth 3, roulette 3;
C5 with 4x4 antialiasing) twice: whole frames at the largest resolution the CPU oracle -- a literal linear scan over
228 / 10 002 shapes per ray -- finishes in about a minute on the GPU box's 16 host threads, and at their FULL quoted
sizes (3840x2160; 7680x4320 x 16 spp) with whole rows / 64-pixel windows of the frame checked against the oracle plus
the whole-frame-equals-its-parts property.  Bar: bit-exact against the strict-math oracle, every counter equal.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import (demo_camera, demo_world, flake_camera, flake_world, random_spheres_camera,
                                  random_spheres_world)
from rustracer_amd.world import flatten_camera, flatten_world

from test_gpu_parity import COUNTERS, assert_bit_equal, gpu_image, gpu_render

pytestmark = pytest.mark.gpu

THREADS = int(os.environ.get("OMP_NUM_THREADS", "0")) or 8


def oracle(world, cam, renderer, w, h, aa=1, mode=O.MATH_STRICT):
    return O.fire_all_rays(flatten_world(world), flatten_camera(cam), renderer.flatten(), w, h, 42, 54, aa,
                           math_mode=mode, n_threads=THREADS)


def test_c2_flat_1280x960_aa4_full_size_bit_exact():
    """BASELINE config 2 as quoted: demo scene, flat renderer, 1280x960, 4x4 antialiasing."""
    w, h, aa = 1280, 960, 4
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.FlatRenderer(world, ra.BLACK)
    ref, rs = oracle(world, cam, rr, w, h, aa)
    g, gs = gpu_image(world, cam, rr, w, h, aa=aa)
    assert_bit_equal(g, ref, "C2 full size")
    assert rs["rays"] == w * h * aa * aa
    for k in COUNTERS:
        assert gs[k] == rs[k], k


def test_c4_flake_textured_at_the_configs_renderer_parameters():
    """BASELINE config 4: sphere-flake with checkered + image pigments, pathtracer n=10 d=3 rr=3 (1 110 rays per pixel)."""
    w, h = 320, 180
    world = flake_world(True)
    cam, rr = flake_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, rs = oracle(world, cam, rr, w, h)
    for schedule in [(), (0,)]:
        g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
        assert_bit_equal(g.cpu().numpy(), ref, f"C4 n=10 d=3 rr=3 schedule={schedule}")
        for k in COUNTERS:
            assert gs[k] == rs[k], (schedule, k, gs[k], rs[k])


def test_c5_random_spheres_aa4_at_the_configs_renderer_parameters():
    """BASELINE config 5: 10 002 shapes, pathtracer n=10 d=3 rr=3, 16 samples per pixel (through the hierarchy)."""
    w, h, aa = 32, 18, 4
    world = random_spheres_world(10000)
    cam, rr = random_spheres_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, rs = oracle(world, cam, rr, w, h, aa)
    g, gs = gpu_render(world, cam, rr, w, h, aa=aa)
    assert_bit_equal(g.cpu().numpy(), ref, "C5 n=10 d=3 rr=3 aa=4")
    for k in COUNTERS:
        assert gs[k] == rs[k], (k, gs[k], rs[k])


def whole_frame_and_parts(world, cam, rr, w, h, aa, n_parts, whole_schedule, parts_schedule):
    """One frame twice on the GPU: whole in one launch, and as n_parts row-tile parts (1-row tiles dealt round-robin)
    assembled by rt_unpack_parts.  Returns (whole image on the device, its stats, assembled image, summed part stats)."""
    import torch
    lib = _abi.lib()
    c, r = flatten_camera(cam), rr.flatten()
    dev = world.device()
    whole, ws = gpu_render(world, cam, rr, w, h, aa=aa, schedule=whole_schedule)
    dev.set_schedule(parts_schedule)
    sizes = [lib.rt_partition_pixels(w, h, C.byref(_abi.rt_partition(p, n_parts, 1, 0))) for p in range(n_parts)]
    assert sum(sizes) == w * h
    stride = max(sizes)
    parts = torch.zeros((n_parts, stride, 3), dtype=torch.float32, device="cuda")
    tot = {k: 0 for k in COUNTERS}
    for p in range(n_parts):
        part = _abi.rt_partition(p, n_parts, 1, 0)
        st = _abi.rt_stats()
        _abi.check(lib.rt_render(dev.handle, C.byref(c), C.byref(r), w, h, 42, 54, aa, C.byref(part), parts[p].data_ptr(),
                                 None, C.byref(st)))
        print(f"  part {p}/{n_parts}: {st.kernel_ms:.0f} ms, {st.rays} rays", flush=True)
        for k in COUNTERS:
            tot[k] += getattr(st, k)
    dev.set_schedule(())
    image = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    _abi.check(lib.rt_unpack_parts(parts.data_ptr(), stride, n_parts, 1, w, h, image.data_ptr(), None))
    torch.cuda.synchronize()
    del parts
    return whole, ws, image, tot


def test_c4_flake_full_size_3840x2160_rows_against_the_oracle():
    """BASELINE config 4 at its quoted size: the textured sphere-flake (228 shapes, checkered + image pigments), pathtracer
    n=10 d=3 rr=3, 3840x2160.  The oracle's literal linear scan cannot shade 8.3 M pixels in the suite's time, but it takes
    a row range: four whole rows spread over the frame (the serial master stream walked up to each row, src/imagetracer.rs:
    61-84; src/world.rs:32-50 over 228 shapes) are compared bit for bit with the same rows of the full-size GPU frame.
    Size-independent property on top: the whole frame in one launch == the frame as 8 row-tile parts under the other
    schedule, every counter equal, every pixel finite."""
    import torch
    w, h = 3840, 2160
    world = flake_world(True)
    cam, rr = flake_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    whole, ws, image, tot = whole_frame_and_parts(world, cam, rr, w, h, 1, 8, (0,), ())
    assert torch.equal(whole.view(torch.int32), image.view(torch.int32)), "C4 full size: parts differ from the whole frame"
    for k in COUNTERS:
        assert tot[k] == ws[k], (k, tot[k], ws[k])
    assert bool(torch.isfinite(whole).all())
    assert ws["samples"] == w * h
    print(f"C4 full size: {ws['rays']} rays, kernel {ws['kernel_ms']:.0f} ms")
    flat, c, r = flatten_world(world), flatten_camera(cam), rr.flatten()
    for row in (0, 641, 1378, 2159):
        ref, rs = O.fire_all_rays(flat, c, r, w, h, 42, 54, 1, rows=(row, row + 1), n_threads=THREADS)
        assert_bit_equal(whole[row:row + 1].cpu().numpy(), ref, f"C4 3840x2160 row {row}")
        assert rs["samples"] == w


def test_c5_random_spheres_full_size_7680x4320_16spp_windows_against_the_oracle():
    """BASELINE config 5 at its quoted size: 10 002 shapes, 7680x4320, 16 samples per pixel, pathtracer n=10 d=3 rr=3 (1.03e11
    rays through the exact-winner hierarchy).  Against the oracle (literal scan of all 10 002 shapes per ray): 64-pixel
    windows spread over the frame -- top (sky), the horizon band, the spheres, the bottom-right corner -- bit for bit.
    Size-independent property: whole frame (one launch) == 8 row-tile parts (short-pass schedule) assembled by
    rt_unpack_parts, counters equal, all finite."""
    import torch
    w, h, aa = 7680, 4320, 4
    world = random_spheres_world(10000)
    cam, rr = random_spheres_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    whole, ws, image, tot = whole_frame_and_parts(world, cam, rr, w, h, aa, 8, (0,), ())
    assert torch.equal(whole.view(torch.int32), image.view(torch.int32)), "C5 full size: parts differ from the whole frame"
    del image
    for k in COUNTERS:
        assert tot[k] == ws[k], (k, tot[k], ws[k])
    assert bool(torch.isfinite(whole).all())
    assert ws["samples"] == w * h * aa * aa
    print(f"C5 full size: {ws['rays']} rays, kernel {ws['kernel_ms']:.0f} ms")
    flat, c, r = flatten_world(world), flatten_camera(cam), rr.flatten()
    for row, col in ((0, 0), (1900, 3808), (2477, 1111), (3200, 6000), (4319, 7616)):
        ref, rs = O.fire_all_rays(flat, c, r, w, h, 42, 54, aa, rows=(row, row + 1), cols=(col, col + 64), n_threads=THREADS)
        assert_bit_equal(whole[row:row + 1, col:col + 64].cpu().numpy(), ref, f"C5 7680x4320x16spp window ({row}, {col})")
        assert rs["samples"] == 64 * aa * aa and rs["rays"] > 0
        print(f"  window ({row}, {col}): oracle {rs['rays']} rays", flush=True)


def libm_outliers(gpu_img, libm_img):
    """Pixels whose radiance differs from the libm-mode oracle by more than 1e-4 relative (north_star's tolerance)."""
    g, o = np.asarray(gpu_img, dtype=np.float64), np.asarray(libm_img, dtype=np.float64)
    tol = 1e-4 * np.maximum(np.abs(g), np.abs(o)) + 1e-9
    bad = (np.abs(g - o) > tol).any(axis=-1)
    return int(bad.sum()), bad.size


def test_headline_frame_against_the_libm_oracle_outlier_fraction():
    """The reference computes sin/cos/atan2/acos with the platform libm (src/material.rs:180, src/shape.rs:102-103),
    whose last ulp Cargo.lock does not pin.  Against THIS box's glibc: every pixel within 1e-4 relative, except those
    where a last-ulp difference flips a discrete decision (hit/miss, roulette) and re-routes the pixel's PCG stream;
    those are counted, never hidden, and must stay a handful."""
    w, h = 1920, 1080
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    libm, ls = oracle(world, cam, rr, w, h, mode=O.MATH_LIBM)
    g, gs = gpu_render(world, cam, rr, w, h, schedule=(0,))
    n_bad, n = libm_outliers(g.cpu().numpy(), libm)
    frac = n_bad / n
    print(f"headline frame vs libm oracle: {n_bad} of {n} pixels beyond 1e-4 relative ({frac:.2e}); rays {gs['rays']} vs {ls['rays']}")
    assert frac <= 2e-5, (n_bad, n)                       # measured: a handful of 2 073 600
    assert abs(gs["rays"] - ls["rays"]) <= 2000, (gs["rays"], ls["rays"])


def test_fire_all_rays_fans_out_over_the_listed_devices(monkeypatch):
    """The drop-in call's internal fan-out, rehearsed on one GPU by listing device 0 several times: N parts (row tiles
    dealt round-robin), each with its own scene copy, stream and staging buffer, must give the whole frame bit for bit."""
    lib = _abi.lib()
    w, h = 200, 113
    world = demo_world()
    cam, rr = demo_camera(w, h), ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    ref, rs = oracle(world, cam, rr, w, h)
    for devices, n_parts in (("0,0,0", 3), ("0", 1), ("0,0,0,0,0,0,0,0", 8)):
        monkeypatch.setenv("RUSTRACER_DEVICES", devices)
        g, gs = gpu_image(world, cam, rr, w, h)
        assert_bit_equal(g, ref, f"fan-out {devices}")
        for k in COUNTERS:
            assert gs[k] == rs[k], (devices, k, gs[k], rs[k])
        ms = (C.c_double * 16)()
        dv = (C.c_int32 * 16)()
        assert lib.rt_last_fire_devices(dv, ms, 16) == n_parts
        assert all(dv[i] == 0 for i in range(n_parts)) and all(ms[i] > 0 for i in range(n_parts))
    # another World through the same cached contexts, then the first one again (aa 2, flat: the image kernel)
    monkeypatch.setenv("RUSTRACER_DEVICES", "0,0")
    other = flake_world(True)
    for wd, cm, r2, aa in ((other, flake_camera(64, 36), ra.FlatRenderer(other, ra.BLACK), 2), (world, cam, rr, 1)):
        ww, hh = (64, 36) if wd is other else (w, h)
        g, gs = gpu_image(wd, cm, r2, ww, hh, aa=aa)
        o, os_ = oracle(wd, cm, r2, ww, hh, aa)
        assert_bit_equal(g, o, "fan-out after a scene change")
        assert gs["rays"] == os_["rays"]
    monkeypatch.delenv("RUSTRACER_DEVICES")
    # rt_fire_release lets go of every cached context; the next call builds them again and renders the same frame
    lib.rt_fire_release()
    lib.rt_fire_release()
    g, gs = gpu_image(world, cam, rr, w, h)
    assert_bit_equal(g, ref, "after rt_fire_release")
    assert gs["rays"] == rs["rays"]
    # a World that differs from the cached one in a single material colour must not be served from the cache
    world2 = demo_world()
    flat2 = flatten_world(world2)
    flat2.materials[0].emitted.c1[0] = 0.25
    c2, r2 = flatten_camera(cam), rr.flatten()
    out2 = np.zeros((h, w, 3), dtype=np.float32)
    _abi.check(lib.rt_fire_all_rays(*flat2.args(), C.byref(c2), C.byref(r2), w, h, 42, 54, 1, O.fptr(out2), None))
    ref2, _ = O.fire_all_rays(flat2, c2, r2, w, h, 42, 54, 1, n_threads=THREADS)
    assert_bit_equal(out2, ref2, "fan-out cache after a one-colour change")
    assert not np.array_equal(out2, ref)
    monkeypatch.setenv("RUSTRACER_NUM_DEVICES", "7")       # more devices than the box has
    out = np.zeros((4, 4, 3), dtype=np.float32)
    st = _abi.rt_stats()
    flat = flatten_world(world)
    c4, r4 = flatten_camera(demo_camera(4, 4)), rr.flatten()
    rc = lib.rt_fire_all_rays(*flat.args(), C.byref(c4), C.byref(r4), 4, 4, 42, 54, 1, O.fptr(out), C.byref(st))
    import torch
    if torch.cuda.device_count() < 7:
        assert rc != 0 and b"RUSTRACER_NUM_DEVICES" in lib.rt_last_error()


def test_rotating_row_blocks_land_in_place_without_unpack():
    """rt_partition.rotate: frame f of a launch renders part (part + f) % n_parts; with tile_rows = height / n_parts
    every part is one contiguous block of rows, written straight to its place in the frame's final image."""
    import torch
    lib = _abi.lib()
    w, h, n_parts, n_frames = 96, 60, 4, 6
    world = demo_world()
    rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    cams = [demo_camera(w, h, angle_deg=7.0 * f) for f in range(n_frames)]        # an animation: every frame differs
    refs = [oracle(world, c, rr, w, h)[0] for c in cams]
    block = h // n_parts
    images = torch.full((n_frames, h, w, 3), float("nan"), dtype=torch.float32, device="cuda")
    carr = (_abi.rt_camera * n_frames)(*[flatten_camera(c) for c in cams])
    r = rr.flatten()
    total_rays = 0
    for p in range(n_parts):
        part = _abi.rt_partition(p, n_parts, block, 1)
        # frame f's block (p + f) % n_parts, in place in the frame's image
        outs = (C.c_void_p * n_frames)(*[images[f, ((p + f) % n_parts) * block].data_ptr() for f in range(n_frames)])
        st = _abi.rt_stats()
        _abi.check(lib.rt_render_frames(world.device().handle, carr, n_frames, C.byref(r), w, h, 42, 54, 1, C.byref(part),
                                        outs, torch.cuda.current_stream().cuda_stream, C.byref(st)))
        total_rays += st.rays
    torch.cuda.synchronize()
    got = images.cpu().numpy()
    for f in range(n_frames):
        assert_bit_equal(got[f], refs[f], f"frame {f} assembled from rotating blocks")
    # unequal parts are refused
    bad = _abi.rt_partition(0, 4, 7, 1)
    outs = (C.c_void_p * 1)(images[0].data_ptr())
    assert lib.rt_render_frames(world.device().handle, carr, 1, C.byref(r), w, h, 42, 54, 1, C.byref(bad), outs, None, None) != 0
    assert b"equal parts" in lib.rt_last_error()


def test_brdf_scatter_ray_units_against_the_oracle():
    """SURVEY §8 a11 / a12 directly: Diffuse (cosine-weighted, ONB from the RAW normal) and Specular scatter_ray on the
    GPU against oracle_scatter_ray, every f32 of the new ray; includes PCG floats of exactly 0 and 1, normals with
    z = -1 (the ONB's singular direction) and non-unit normals."""
    lib = _abi.lib()
    rng = np.random.default_rng(11)
    n = 4096
    rf = (rng.integers(0, 2**32, (n, 2), dtype=np.uint64).astype(np.float32) * np.float32(2.0**-32)).astype(np.float32)
    rf[0], rf[1], rf[2] = (0.0, 0.0), (1.0, 1.0), (0.0, 1.0)
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    normal[: n // 2] /= np.linalg.norm(normal[: n // 2], axis=1, keepdims=True).astype(np.float32)   # half unit, half raw
    normal[3], normal[4], normal[5] = (0, 0, -1), (0, 0, 1), (1e-8, 0, -1)
    incoming = rng.normal(size=(n, 3)).astype(np.float32)
    point = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    for brdf in (_abi.RT_BRDF_DIFFUSE, _abi.RT_BRDF_SPECULAR):
        got = (_abi.rt_ray * n)()
        _abi.check(lib.rt_brdf_scatter_ray(brdf, O.fptr(rf), O.fptr(incoming), O.fptr(point), O.fptr(normal), 2, n, got))
        for i in range(n):
            ref = _abi.rt_ray()
            O.lib().oracle_scatter_ray(brdf, float(rf[i, 0]), float(rf[i, 1]), O.fptr(incoming[i]), O.fptr(point[i]),
                                       O.fptr(normal[i]), 2, O.MATH_STRICT, C.byref(ref))
            a = np.array(list(got[i].origin) + list(got[i].dir) + [got[i].tmin, got[i].tmax], dtype=np.float32)
            b = np.array(list(ref.origin) + list(ref.dir) + [ref.tmin, ref.tmax], dtype=np.float32)
            assert_bit_equal(a, b, f"scatter_ray brdf={brdf} item {i}")
            assert got[i].depth == ref.depth == 2


def test_in_place_parts_assemble_in_one_image():
    """rt_partition.in_place: every part stores its pixels at their global position in ONE full-size image (what each
    GPU's kernel does to rank 0's image across xGMI): 1-row tiles, 3-row tiles, and rotating blocks over several frames."""
    import torch
    lib = _abi.lib()
    w, h = 120, 64
    world = demo_world()
    rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    r = rr.flatten()
    cam = demo_camera(w, h)
    ref, rs = oracle(world, cam, rr, w, h)
    c = flatten_camera(cam)
    for n_parts, tile_rows in ((4, 1), (3, 3), (8, 8)):
        image = torch.full((h, w, 3), float("nan"), dtype=torch.float32, device="cuda")
        rays = 0
        for p in range(n_parts):
            part = _abi.rt_partition(p, n_parts, tile_rows, 0, 1)
            st = _abi.rt_stats()
            _abi.check(lib.rt_render(world.device().handle, C.byref(c), C.byref(r), w, h, 42, 54, 1, C.byref(part),
                                     image.data_ptr(), torch.cuda.current_stream().cuda_stream, C.byref(st)))
            rays += st.rays
        assert_bit_equal(image.cpu().numpy(), ref, f"in place {n_parts} parts x {tile_rows} rows")
        assert rays == rs["rays"]
    # flat renderer (the image kernel), antialiased, in place
    fl = ra.FlatRenderer(world, ra.BLACK)
    ref2, _ = oracle(world, cam, fl, w, h, aa=2)
    image = torch.full((h, w, 3), float("nan"), dtype=torch.float32, device="cuda")
    f = fl.flatten()
    for p in range(2):
        part = _abi.rt_partition(p, 2, 1, 0, 1)
        _abi.check(lib.rt_render(world.device().handle, C.byref(c), C.byref(f), w, h, 42, 54, 2, C.byref(part),
                                 image.data_ptr(), torch.cuda.current_stream().cuda_stream, None))
    torch.cuda.synchronize()
    assert_bit_equal(image.cpu().numpy(), ref2, "flat, in place")
    # rotating blocks of 4 frames, in place
    n_parts, n_frames = 4, 4
    images = torch.full((n_frames, h, w, 3), float("nan"), dtype=torch.float32, device="cuda")
    carr = (_abi.rt_camera * n_frames)(*([c] * n_frames))
    outs = (C.c_void_p * n_frames)(*[images[k].data_ptr() for k in range(n_frames)])
    for p in range(n_parts):
        part = _abi.rt_partition(p, n_parts, h // n_parts, 1, 1)
        _abi.check(lib.rt_render_frames(world.device().handle, carr, n_frames, C.byref(r), w, h, 42, 54, 1, C.byref(part), outs,
                                        torch.cuda.current_stream().cuda_stream, None))
    torch.cuda.synchronize()
    for k in range(n_frames):
        assert_bit_equal(images[k].cpu().numpy(), ref, f"rotating blocks in place, frame {k}")


def test_diagonal_inverse_specialisation_and_its_guard():
    """Scenes made of scalings and translations take the sparse form of the closest-hit arithmetic (rt_device.h:
    diag_ray_ok); it must equal the reference bit for bit, and rays it cannot prove that for -- a direction component of
    exactly zero, far-away origins -- must fall back to the full products.  Negative and extreme scale factors included."""
    world = ra.World()
    sky = ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.UniformPigment((1.0, 0.9, 0.5)))
    ground = ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment((0.3, 0.5, 0.1), (0.1, 0.2, 0.5), 4)), ra.UniformPigment(ra.BLACK))
    blue = ra.Material(ra.DiffuseBRDF(ra.UniformPigment((0.3, 0.4, 0.8))), ra.UniformPigment(ra.BLACK))
    mirror = ra.Material(ra.SpecularBRDF(ra.UniformPigment((0.6, 0.2, 0.3))), ra.UniformPigment(ra.BLACK))
    world.add(ra.Sphere(ra.scaling((150.0, 150.0, 150.0)), sky))
    world.add(ra.Plane(ra.translation((0.0, 0.0, -0.5)), ground))
    world.add(ra.Sphere(ra.translation((0.0, 0.0, 0.0)) * ra.scaling((-0.7, 0.4, 1.3)), blue))        # negative scale, ellipsoid
    world.add(ra.Sphere(ra.translation((1.0, 1.5, 0.2)) * ra.scaling((0.5, 0.5, 0.5)), mirror))
    world.add(ra.Sphere(ra.translation((0.5, -1.5, 0.0)) * ra.scaling((1e-3, 2.0, 0.25)), blue))      # thin
    rr = ra.PathTracer(world, ra.BLACK, 5, 3, 2)
    w, h = 96, 64
    cams = [ra.PerspectiveCamera(1.0, 1.5, ra.translation((-3.0, 0.0, 0.4))),                # generic rays: the sparse form
            ra.OrthogonalCamera(1.5, ra.translation((-3.0, 0.0, 0.0))),                      # d = (1, 0, 0): zero components
            ra.PerspectiveCamera(1.0, 1.5, ra.translation((-2.0e18, 0.0, 0.4))),             # origin beyond the guard's bound
            ra.PerspectiveCamera(1.0, 1.5, ra.rotation_z(0.3) * ra.translation((-3.0, 0.0, 0.4)))]
    for cam in cams:
        ref, rs = oracle(world, cam, rr, w, h)
        for schedule in [(0,), ()]:
            g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
            assert_bit_equal(g.cpu().numpy(), ref, f"diagonal scene, {type(cam).__name__}, schedule={schedule}")
            for k in COUNTERS:
                assert gs[k] == rs[k], (k, gs[k], rs[k])
        fl = ra.FlatRenderer(world, ra.BLACK)
        ref2, _ = oracle(world, cam, fl, w, h, aa=2)
        g2, _ = gpu_image(world, cam, fl, w, h, aa=2)
        assert_bit_equal(g2, ref2, f"diagonal scene, flat, {type(cam).__name__}")


@pytest.mark.parametrize("n,d,rr_limit", [(4, 3, 0), (10, 2, 1), (3, 4, 2), (10, 3, 3), (70, 1, 0)])
def test_mirror_children_runs_and_their_exactness_conditions(n, d, rr_limit, monkeypatch):
    """The children of a mirror hit trace one ray; where no draw can change their value the throughput body folds the rest
    of them at once and jumps the pixel's PCG over their draws (k_pass FAST, `RT_MIRROR_RUNS`).  That is only exact if
    `emitted + 0` leaves the emitted colour alone, so emitted colours of -0, NaN and inf -- with and without a roulette
    draw at that depth -- must take the one-by-one path; rays that leave the scene return a background with a -0 in it;
    70 children exceed the jump table (3 * 69 draws) and fall back too."""
    monkeypatch.setenv("RUSTRACER_MAX_TEAM", "1")            # small frames through the throughput body
    w, h = 80, 56
    mirror = ra.Material(ra.SpecularBRDF(ra.UniformPigment((0.7, 0.6, 0.9))), ra.UniformPigment(ra.BLACK))
    grey = ra.Material(ra.DiffuseBRDF(ra.UniformPigment((0.5, 0.5, 0.5))), ra.UniformPigment(ra.BLACK))
    for emitted, with_sky, bg in [((1.0, 0.9, 0.5), True, ra.BLACK), ((-0.0, 0.5, 1.0), True, ra.BLACK),
                                  ((float("nan"), 0.2, -0.0), True, ra.BLACK), ((float("inf"), 0.0, 0.3), True, ra.BLACK),
                                  (None, False, (0.1, -0.0, 0.3))]:
        world = ra.World()
        if with_sky:
            world.add(ra.Sphere(ra.scaling((40.0, 40.0, 40.0)),
                                ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.UniformPigment(emitted))))
        world.add(ra.Plane(ra.translation((0.0, 0.0, -1.0)), grey))
        world.add(ra.Sphere(ra.translation((0.3, -0.4, 0.0)), mirror))
        world.add(ra.Sphere(ra.translation((0.8, 1.4, -0.4)) * ra.scaling((0.6, 0.6, 0.6)), mirror))
        cam = ra.PerspectiveCamera(1.0, w / h, ra.translation((-2.5, 0.0, 0.3)))
        rr = ra.PathTracer(world, bg, n, d, rr_limit)
        ref, rs = oracle(world, cam, rr, w, h)
        for schedule in [(0,), ()]:
            g, gs = gpu_render(world, cam, rr, w, h, schedule=schedule)
            assert_bit_equal(g.cpu().numpy(), ref, f"mirror runs, emitted={emitted}, bg={bg}, schedule={schedule}")
            for k in COUNTERS:
                assert gs[k] == rs[k], (emitted, schedule, k, gs[k], rs[k])


@pytest.mark.parametrize("n,d,rr_limit", [(10, 3, 3), (10, 3, 5), (10, 2, 0), (3, 3, 1), (30, 2, 2), (2, 4, 2), (1, 3, 0)])
@pytest.mark.parametrize("helpers", ["1", "2", "3"])
def test_drain_helpers_and_same_round_mirror_siblings_are_exact(n, d, rr_limit, helpers, monkeypatch):
    """Once the queue is dry the throughput body (k_pass FAST) (a) resolves all remaining siblings of a mirror hit's leaf child
    in one round and (b) lets free lanes trace later siblings of another lane's last-level diffuse hit at speculated PCG
    offsets (kernel variants 2 and 3; 3 also mixes the queue's last hand-outs; 1 is the plain body: `RUSTRACER_FAST_VARIANT`).  All must reproduce the
    sequential recursion bit for bit: with a roulette draw at the last level (rr_limit <= max_depth) and without, with
    rays that leave the scene (no sky: they draw nothing, which takes a sibling off the speculated grid), with more
    children than the jump tables cover (n = 30), a single child (nothing to speculate) and every counter equal."""
    monkeypatch.setenv("RUSTRACER_MAX_TEAM", "1")            # small frames through the throughput body
    monkeypatch.setenv("RUSTRACER_FAST_VARIANT", helpers)
    w, h = 96, 64
    mirror = ra.Material(ra.SpecularBRDF(ra.UniformPigment((0.7, 0.6, 0.9))), ra.UniformPigment(ra.BLACK))
    grey = ra.Material(ra.DiffuseBRDF(ra.CheckeredPigment((0.5, 0.5, 0.5), (0.9, 0.2, 0.1), 6)), ra.UniformPigment(ra.BLACK))
    blue = ra.Material(ra.DiffuseBRDF(ra.UniformPigment((0.3, 0.4, 0.8))), ra.UniformPigment((0.05, 0.0, 0.1)))
    for with_sky, bg in [(True, ra.BLACK), (False, (0.2, 0.3, 0.9))]:
        world = ra.World()
        if with_sky:
            world.add(ra.Sphere(ra.scaling((40.0, 40.0, 40.0)),
                                ra.Material(ra.DiffuseBRDF(ra.UniformPigment(ra.BLACK)), ra.UniformPigment((1.0, 0.9, 0.5)))))
        world.add(ra.Plane(ra.translation((0.0, 0.0, -1.0)), grey))
        world.add(ra.Sphere(ra.translation((0.3, -0.6, -0.2)), blue))
        world.add(ra.Sphere(ra.translation((0.8, 1.4, -0.4)) * ra.scaling((0.6, 0.6, 0.6)), mirror))
        cam = ra.PerspectiveCamera(1.0, w / h, ra.translation((-2.5, 0.0, 0.3)))
        rr = ra.PathTracer(world, bg, n, d, rr_limit)
        ref, rs = oracle(world, cam, rr, w, h)
        g, gs = gpu_render(world, cam, rr, w, h, schedule=(0,))
        assert_bit_equal(g.cpu().numpy(), ref, f"kernel variant {helpers}, sky={with_sky}, n={n} d={d} rr={rr_limit}")
        for k in COUNTERS:
            assert gs[k] == rs[k], (with_sky, k, gs[k], rs[k])
    if (n, d, rr_limit) == (10, 3, 3):
        # antialiased: a lane between two samples of its pixel holds state (master PCG, accumulator) and is no free lane
        ref, rs = oracle(world, cam, rr, 64, 40, aa=3)
        g, gs = gpu_render(world, cam, rr, 64, 40, aa=3, schedule=(0,))
        assert_bit_equal(g.cpu().numpy(), ref, f"kernel variant {helpers}, aa=3")
        for k in COUNTERS:
            assert gs[k] == rs[k], ("aa=3", k, gs[k], rs[k])
