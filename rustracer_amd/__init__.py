"""rustracer_amd — MI355X-native render core for andros21/rustracer's per-pixel render loop.

Host mirror of the reference's interface for the hot path (same names, argument meaning and
error behaviour) over the C-ABI in include/rustracer_amd.h; the compute runs in hand-written
gfx950 HIP kernels (csrc/).  There is no CPU fallback.
"""
from .camera import OrthogonalCamera, PerspectiveCamera
from .hdrimage import HdrImage, HdrImageErr
from .imagetracer import ImageTracer
from .material import (BLACK, WHITE, CheckeredPigment, DiffuseBRDF, ImagePigment, Material,
                       SpecularBRDF, UniformPigment)
from .random import Pcg
from .ray import Ray
from .render import DummyRenderer, FlatRenderer, OnOffRenderer, PathTracer
from .shape import Plane, Sphere
from .transformation import (Transformation, rotation_x, rotation_y, rotation_z, scaling,
                             to_radians, translation)
from .world import DeviceScene, World, flatten_camera, flatten_world

__all__ = [n for n in dir() if not n.startswith("_")]
