"""Host mirror of src/camera.rs (parameters; Camera::fire_ray runs on the GPU)."""
from dataclasses import dataclass, field

from .transformation import Transformation


@dataclass
class OrthogonalCamera:
    """src/camera.rs:25-46."""
    aspect_ratio: float = 1.0
    transformation: Transformation = field(default_factory=Transformation)


@dataclass
class PerspectiveCamera:
    """src/camera.rs:82-110."""
    distance: float = 1.0
    aspect_ratio: float = 1.0
    transformation: Transformation = field(default_factory=Transformation)
