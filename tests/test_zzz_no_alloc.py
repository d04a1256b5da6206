"""rt_render_noalloc (include/rustracer_amd.h): the allocation-free variant of rt_render, through the C-ABI.

Its output must equal rt_render's bit for bit (same kernels; only the workspace differs: the file-scope device
scratch instead of the scene's allocated work slots), on two calls in a row, the second with a smaller input,
so that whatever the first call left in the shared scratch is seen to be cleared.  Collected last (file name):
a failure here must not hide the other GPU tests from `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest

import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera

pytestmark = pytest.mark.gpu


def _both(world, cam, renderer, w, h, aa=1, part=None):
    """(rt_render image, rt_render_noalloc image) of the same call, as uint32 views."""
    import torch
    lib = _abi.lib()
    dev = world.device()
    c, r = flatten_camera(cam), renderer.flatten()
    n_pix = lib.rt_partition_pixels(w, h, C.byref(part)) if part is not None else w * h
    a = torch.full((n_pix, 3), float("nan"), dtype=torch.float32, device="cuda")
    b = torch.full((n_pix, 3), float("nan"), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    p = C.byref(part) if part is not None else None
    _abi.check(lib.rt_render(dev.handle, C.byref(c), C.byref(r), w, h, 42, 54, aa, p, a.data_ptr(), s, None))
    _abi.check(lib.rt_render_noalloc(dev.handle, C.byref(c), C.byref(r), w, h, 42, 54, aa, p, b.data_ptr(), s))
    torch.cuda.synchronize()
    return a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)


def test_noalloc_equals_rt_render_two_calls_second_smaller():
    _abi.check(_abi.lib().rt_init(0))
    world = demo_world()
    rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3)
    for w, h in ((320, 180), (96, 54)):          # the second call is the smaller one
        a, b = _both(world, demo_camera(w, h), rr, w, h)
        assert not np.isnan(a.view(np.float32)).any()
        assert np.array_equal(a, b), f"{w}x{h}: {int((a != b).sum())} words differ"


def test_noalloc_other_renderers_partition_and_deep_recursion():
    world = demo_world()
    w, h = 160, 90
    cam = demo_camera(w, h)
    # recursion too deep for LDS: the stack lives in the scratch (after a shallow call, then a smaller deep one)
    for rr, (ww, hh), aa in ((ra.PathTracer(world, ra.BLACK, 2, 7, 3), (w, h), 1),
                             (ra.PathTracer(world, ra.BLACK, 1, 40, 2), (64, 36), 2),
                             (ra.FlatRenderer(world, ra.BLACK), (w, h), 2),
                             (ra.OnOffRenderer(world), (w, h), 1)):
        a, b = _both(world, demo_camera(ww, hh), rr, ww, hh, aa=aa)
        assert np.array_equal(a, b), (type(rr).__name__, ww, hh)
    part = _abi.rt_partition(1, 3, 2)
    a, b = _both(world, cam, ra.PathTracer(world, ra.BLACK, 10, 3, 3), w, h, part=part)
    assert np.array_equal(a, b)


def test_noalloc_rejects_what_the_scratch_cannot_hold():
    import torch
    lib = _abi.lib()
    world = demo_world()
    dev = world.device()
    w, h = 32, 18
    c = flatten_camera(demo_camera(w, h))
    r = ra.PathTracer(world, ra.BLACK, 1, _abi.RT_NOALLOC_MAX_DEPTH + 1, 3).flatten()
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    rc = lib.rt_render_noalloc(dev.handle, C.byref(c), C.byref(r), w, h, 42, 54, 1, None, out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    assert b"RT_NOALLOC_MAX_DEPTH" in lib.rt_last_error()
    # the largest depth it accepts still renders what rt_render renders
    rr = ra.PathTracer(world, ra.BLACK, 1, _abi.RT_NOALLOC_MAX_DEPTH, 3)
    a, b = _both(world, demo_camera(w, h), rr, w, h)
    assert np.array_equal(a, b)
