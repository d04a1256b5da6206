"""Host mirror of src/render.rs: renderer parameter objects; Solve::solve runs on the GPU."""
import ctypes as C
from dataclasses import dataclass

from . import _abi
from .material import BLACK, WHITE
from .world import World, to_rt_ray


class _Renderer:
    algorithm = _abi.RT_ALGO_DUMMY
    world = None

    def flatten(self):
        r = _abi.rt_renderer()
        r.algorithm = self.algorithm
        r.bg[:] = [float(c) for c in getattr(self, "bg_color", BLACK)]
        r.fg[:] = [float(c) for c in getattr(self, "fg_color", WHITE)]
        r.num_of_rays = int(getattr(self, "num_of_rays", 0))
        r.max_depth = int(getattr(self, "max_depth", 0))
        r.rr_limit = int(getattr(self, "russian_roulette_limit", 0))
        return r

    def solve(self, ray, pcg):
        """Solve::solve (src/render.rs:18-20, :195-205) for one (ray, pcg) on the GPU.

        `pcg` is an object with integer `.state` / `.inc`; it is advanced in place."""
        world = self.world if self.world is not None else World()
        rays = (_abi.rt_ray * 1)(to_rt_ray(ray))
        pcgs = (_abi.rt_pcg * 1)()
        pcgs[0].state, pcgs[0].inc = pcg.state, pcg.inc
        out = (C.c_float * 3)()
        rr = self.flatten()
        _abi.check(_abi.lib().rt_renderer_solve(world.device().handle, C.byref(rr), rays, pcgs, 1, out))
        pcg.state, pcg.inc = pcgs[0].state, pcgs[0].inc
        return (out[0], out[1], out[2])


@dataclass
class OnOffRenderer(_Renderer):
    """src/render.rs:26-67."""
    world: World
    bg_color: tuple = BLACK
    fg_color: tuple = WHITE
    algorithm = _abi.RT_ALGO_ONOFF


@dataclass
class FlatRenderer(_Renderer):
    """src/render.rs:50-89."""
    world: World
    bg_color: tuple = BLACK
    algorithm = _abi.RT_ALGO_FLAT


@dataclass
class PathTracer(_Renderer):
    """src/render.rs:95-173."""
    world: World
    bg_color: tuple = BLACK
    num_of_rays: int = 10
    max_depth: int = 3
    russian_roulette_limit: int = 3
    algorithm = _abi.RT_ALGO_PATHTRACER


@dataclass
class DummyRenderer(_Renderer):
    """src/render.rs:178-185."""
    world: World = None
    algorithm = _abi.RT_ALGO_DUMMY
