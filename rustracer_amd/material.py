"""Host mirror of src/material.rs: pigments, BRDFs, Material (parameters only; evaluation runs on
the GPU: csrc/rt_device.h pigment_color / *_scatter_dir)."""
import math
from dataclasses import dataclass, field
from typing import Tuple

Color = Tuple[float, float, float]
BLACK: Color = (0.0, 0.0, 0.0)
WHITE: Color = (1.0, 1.0, 1.0)


@dataclass
class UniformPigment:
    """src/material.rs:27-36."""
    color: Color = BLACK


@dataclass
class CheckeredPigment:
    """src/material.rs:77-96."""
    color1: Color = BLACK
    color2: Color = WHITE
    steps: int = 10


@dataclass
class ImagePigment:
    """src/material.rs:42-70; hdr_img is a rustracer_amd.hdrimage.HdrImage."""
    hdr_img: object = None


@dataclass
class DiffuseBRDF:
    """src/material.rs:136-186."""
    pigment: object = field(default_factory=lambda: UniformPigment(WHITE))


@dataclass
class SpecularBRDF:
    """src/material.rs:190-246 (threshold_angle_rad is only read by Eval::eval, never by a renderer)."""
    pigment: object = field(default_factory=lambda: UniformPigment(WHITE))
    threshold_angle_rad: float = math.pi / 1800.0


@dataclass
class Material:
    """src/material.rs:298-312."""
    brdf: object = field(default_factory=DiffuseBRDF)
    emitted_radiance: object = field(default_factory=UniformPigment)
