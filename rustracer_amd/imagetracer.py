"""Host mirror of src/imagetracer.rs: ImageTracer::fire_all_rays through the C-ABI."""
import ctypes as C

import numpy as np

from . import _abi
from .world import World, flatten_camera


class ImageTracer:
    """src/imagetracer.rs:15-42."""

    def __init__(self, image, camera):
        self.image = image
        self.camera = camera
        self.last_stats = None

    def fire_all_rays(self, renderer, init_state, init_seq, antialiasing_level):
        """ImageTracer::fire_all_rays (src/imagetracer.rs:105-125): blocking, fills self.image."""
        world = renderer.world if renderer.world is not None else World()
        flat = world.flatten()
        cam = flatten_camera(self.camera)
        rr = renderer.flatten()
        w, h = self.image.width, self.image.height
        out = np.empty((h, w, 3), dtype=np.float32)
        stats = _abi.rt_stats()
        _abi.check(_abi.lib().rt_fire_all_rays(
            *flat.args(), C.byref(cam), C.byref(rr), w, h, init_state, init_seq,
            antialiasing_level, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(stats)))
        self.last_stats = stats.as_dict()
        self.image.set_pixels(out)
