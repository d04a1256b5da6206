"""Scene inputs of the BASELINE configs, built in code.

demo_world / demo_camera restate the constants of the reference's `demo` subcommand
(src/main.rs:136-213), which is the same scene as examples/demo.yml.
"""
import numpy as np

from .camera import OrthogonalCamera, PerspectiveCamera
from .material import (BLACK, CheckeredPigment, DiffuseBRDF, Material, SpecularBRDF, UniformPigment)
from .shape import Plane, Sphere
from .transformation import Transformation, rotation_z, scaling, to_radians, translation
from .world import World

F = np.float32


def demo_world():
    """src/main.rs:136-196."""
    sky = Material(DiffuseBRDF(UniformPigment(BLACK)), UniformPigment((1.0, 0.9, 0.5)))
    ground = Material(DiffuseBRDF(CheckeredPigment((0.3, 0.5, 0.1), (0.1, 0.2, 0.5), 10)), UniformPigment(BLACK))
    sphere = Material(DiffuseBRDF(UniformPigment((0.3, 0.4, 0.8))), UniformPigment(BLACK))
    mirror = Material(SpecularBRDF(UniformPigment((0.6, 0.2, 0.3))), UniformPigment(BLACK))
    world = World()
    world.add(Sphere(translation((0.0, 0.0, 0.4)) * scaling((200.0, 200.0, 200.0)), sky))
    world.add(Plane(Transformation(), ground))
    world.add(Sphere(translation((0.0, 0.0, 0.1)), sphere))
    world.add(Sphere(translation((1.0, 2.5, 0.0)), mirror))
    return world


def demo_camera(width, height, angle_deg=0.0, orthogonal=False):
    """src/main.rs:197-213: rotation_z(angle+230 deg) * translation(-3.5, 0, 0.7), aspect = w/h in f32."""
    tr = rotation_z(to_radians(F(F(angle_deg) + F(230.0)))) * translation((-3.5, 0.0, 0.7))
    aspect = float(F(F(width) / F(height)))
    if orthogonal:
        return OrthogonalCamera(aspect, tr)
    return PerspectiveCamera(1.0, aspect, tr)
