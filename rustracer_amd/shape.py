"""Host mirror of src/shape.rs: Sphere / Plane {transformation, material}."""
from dataclasses import dataclass, field

from .material import Material
from .transformation import Transformation


@dataclass
class Sphere:
    """src/shape.rs:61-80."""
    transformation: Transformation = field(default_factory=Transformation)
    material: Material = field(default_factory=Material)


@dataclass
class Plane:
    """src/shape.rs:148-167."""
    transformation: Transformation = field(default_factory=Transformation)
    material: Material = field(default_factory=Material)
