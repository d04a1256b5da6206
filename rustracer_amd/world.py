"""Host mirror of src/world.rs: World = ordered list of shapes, flattened to the C-ABI tables.

`World.ray_intersection` runs on the GPU through rt_world_ray_intersection.
"""
import ctypes as C

import numpy as np

from . import _abi
from .camera import OrthogonalCamera, PerspectiveCamera
from .material import (CheckeredPigment, DiffuseBRDF, ImagePigment, Material, SpecularBRDF,
                       UniformPigment)
from .ray import Ray
from .shape import Plane, Sphere


class FlatScene:
    """The caller-owned, read-only tables of include/rustracer_amd.h for one World."""

    def __init__(self, shapes, materials, textures, keepalive):
        self.shapes, self.materials, self.textures = shapes, materials, textures
        self.n_shapes, self.n_materials, self.n_textures = len(shapes), len(materials), len(textures)
        self._keepalive = keepalive

    def args(self):
        return (self.shapes, self.n_shapes, self.materials, self.n_materials, self.textures, self.n_textures)


def _fill_pigment(dst, pig, textures, tex_index, keepalive):
    if isinstance(pig, UniformPigment):
        dst.kind = _abi.RT_PIGMENT_UNIFORM
        dst.c1[:] = [float(c) for c in pig.color]
    elif isinstance(pig, CheckeredPigment):
        dst.kind = _abi.RT_PIGMENT_CHECKERED
        dst.c1[:] = [float(c) for c in pig.color1]
        dst.c2[:] = [float(c) for c in pig.color2]
        dst.steps = int(pig.steps)
    elif isinstance(pig, ImagePigment):
        dst.kind = _abi.RT_PIGMENT_IMAGE
        key = id(pig.hdr_img)
        if key not in tex_index:
            px = np.ascontiguousarray(pig.hdr_img.pixels, dtype=np.float32)
            keepalive.append(px)
            t = _abi.rt_texture()
            t.w, t.h = pig.hdr_img.width, pig.hdr_img.height
            t.rgb = px.ctypes.data_as(C.POINTER(C.c_float))
            tex_index[key] = len(textures)
            textures.append(t)
        dst.texture = tex_index[key]
    else:
        raise TypeError(f"unknown pigment {pig!r}")


def flatten_world(world):
    """World -> rt_shape[] / rt_material[] / rt_texture[]; shape order preserved (world.rs:38)."""
    keepalive, textures, tex_index = [], [], {}
    mats, mat_index = [], {}
    shapes = (_abi.rt_shape * max(1, len(world.shapes)))()
    for i, s in enumerate(world.shapes):
        if isinstance(s, Sphere):
            shapes[i].kind = _abi.RT_SHAPE_SPHERE
        elif isinstance(s, Plane):
            shapes[i].kind = _abi.RT_SHAPE_PLANE
        else:
            raise TypeError(f"unknown shape {s!r}")
        key = id(s.material)
        if key not in mat_index:
            m = _abi.rt_material()
            if isinstance(s.material.brdf, DiffuseBRDF):
                m.brdf = _abi.RT_BRDF_DIFFUSE
            elif isinstance(s.material.brdf, SpecularBRDF):
                m.brdf = _abi.RT_BRDF_SPECULAR
            else:
                raise TypeError(f"unknown BRDF {s.material.brdf!r}")
            _fill_pigment(m.brdf_pigment, s.material.brdf.pigment, textures, tex_index, keepalive)
            _fill_pigment(m.emitted, s.material.emitted_radiance, textures, tex_index, keepalive)
            mat_index[key] = len(mats)
            mats.append(m)
        shapes[i].material = mat_index[key]
        shapes[i].m[:] = [float(v) for v in s.transformation.m.reshape(-1)]
        shapes[i].invm[:] = [float(v) for v in s.transformation.invm.reshape(-1)]
    materials = (_abi.rt_material * max(1, len(mats)))(*mats)
    tex_arr = (_abi.rt_texture * max(1, len(textures)))(*textures)
    fs = FlatScene(shapes, materials, tex_arr, keepalive)
    fs.n_shapes, fs.n_materials, fs.n_textures = len(world.shapes), len(mats), len(textures)
    return fs


def flatten_camera(camera):
    c = _abi.rt_camera()
    if isinstance(camera, OrthogonalCamera):
        c.kind, c.distance = _abi.RT_CAMERA_ORTHOGONAL, 1.0
    elif isinstance(camera, PerspectiveCamera):
        c.kind, c.distance = _abi.RT_CAMERA_PERSPECTIVE, float(camera.distance)
    else:
        raise TypeError(f"unknown camera {camera!r}")
    c.aspect = float(camera.aspect_ratio)
    c.m[:] = [float(v) for v in camera.transformation.m.reshape(-1)]
    return c


def to_rt_ray(ray):
    r = _abi.rt_ray()
    r.origin[:] = [float(c) for c in ray.origin]
    r.dir[:] = [float(c) for c in ray.dir]
    r.tmin, r.tmax, r.depth = float(ray.tmin), float(ray.tmax), int(ray.depth)
    return r


class DeviceScene:
    """World resident in HBM (rt_scene_create / rt_scene_destroy)."""

    def __init__(self, flat):
        self.flat = flat
        self.handle = _abi.lib().rt_scene_create(*flat.args())
        if not self.handle:
            raise _abi.RtError(_abi.lib().rt_last_error().decode())

    def set_schedule(self, budgets=()):
        """rt_scene_set_schedule: () = default passes, (0,) = one pass, else per-pass ray budgets."""
        arr = (C.c_uint32 * max(1, len(budgets)))(*budgets)
        _abi.check(_abi.lib().rt_scene_set_schedule(self.handle, arr, len(budgets)))

    def close(self):
        if self.handle:
            _abi.lib().rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class World:
    """src/world.rs:16-51."""

    def __init__(self):
        self.shapes = []
        self._device = None

    def add(self, shape):
        self.shapes.append(shape)
        self._device = None

    def flatten(self):
        return flatten_world(self)

    def device(self):
        if self._device is None:
            self._device = DeviceScene(self.flatten())
        return self._device

    def ray_intersection(self, ray):
        """World::ray_intersection (src/world.rs:32-50) on the GPU; returns rt_hit or None."""
        hits = self.ray_intersections([ray])
        return hits[0] if hits[0].hit else None

    def ray_intersections(self, rays):
        n = len(rays)
        arr = (_abi.rt_ray * max(1, n))(*[to_rt_ray(r) if isinstance(r, Ray) else r for r in rays])
        hits = (_abi.rt_hit * max(1, n))()
        _abi.check(_abi.lib().rt_world_ray_intersection(self.device().handle, arr, n, hits))
        return list(hits)[:n]
