"""Host mirror of src/ray.rs."""
import math
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class Ray:
    """src/ray.rs:8-21; defaults from Ray::default (src/ray.rs:46-54)."""
    origin: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    dir: Tuple[float, float, float] = (1.0, 0.0, 0.0)
    tmin: float = 1e-5
    tmax: float = math.inf
    depth: int = 0
