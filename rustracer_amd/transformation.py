"""Host mirror of src/transformation.rs: 4x4 f32 Transformation {m, invm} and its builders.

Host-side, runs once per scene; produces the row-major matrices the C-ABI takes
(include/rustracer_amd.h: rt_shape.m / .invm, rt_camera.m).  All arithmetic is f32 in the
reference's operand order so that the matrices equal the reference's bit for bit (up to the
platform libm's sin/cos, which the reference does not pin either).
"""
import math

import numpy as np

F = np.float32
_PI32 = F(math.pi)
_RADS_PER_DEG = F(_PI32 / F(180.0))  # f32::to_radians multiplies by this constant


def to_radians(deg):
    """f32::to_radians (used by src/scene.rs:885-895, :1202 and src/main.rs:197)."""
    return F(F(deg) * _RADS_PER_DEG)


def _matmul(a, b):
    """Matrix * Matrix (src/transformation.rs:64-80): m[i][j] = 0 + a[i][0]*b[0][j] + ... in f32."""
    acc = np.zeros((4, 4), dtype=F)
    for k in range(4):
        acc = (acc + np.outer(a[:, k], b[k, :]).astype(F)).astype(F)
    return acc


class Transformation:
    """src/transformation.rs:87-131."""

    __slots__ = ("m", "invm")

    def __init__(self, m=None, invm=None):
        self.m = np.eye(4, dtype=F) if m is None else np.asarray(m, dtype=F).reshape(4, 4).copy()
        self.invm = np.eye(4, dtype=F) if invm is None else np.asarray(invm, dtype=F).reshape(4, 4).copy()

    def inverse(self):
        return Transformation(self.invm, self.m)

    def is_consistent(self, eps=1e-4):
        return bool(np.all(np.abs(_matmul(self.m, self.invm) - np.eye(4, dtype=F)) < eps))

    def is_close(self, other, eps=1e-4):
        return bool(np.all(np.abs(self.m - other.m) < eps) and np.all(np.abs(self.invm - other.invm) < eps))

    def __mul__(self, rhs):
        if isinstance(rhs, Transformation):
            # m = self.m * rhs.m ; invm = rhs.invm * self.invm (src/transformation.rs:122-131)
            return Transformation(_matmul(self.m, rhs.m), _matmul(rhs.invm, self.invm))
        return NotImplemented

    # T * Point / Vector / Normal (src/transformation.rs:133-195), f32, reference operand order
    def apply_point(self, p):
        m = self.m
        x, y, z = F(p[0]), F(p[1]), F(p[2])
        np_ = [F(F(F(m[i, 0] * x + m[i, 1] * y) + m[i, 2] * z) + m[i, 3]) for i in range(3)]
        w = F(F(F(x * m[3, 0] + y * m[3, 1]) + z * m[3, 2]) + m[3, 3])
        if w == F(1.0):
            return tuple(np_)
        return tuple(F(c / w) for c in np_)

    def apply_vector(self, v):
        m = self.m
        x, y, z = F(v[0]), F(v[1]), F(v[2])
        return tuple(F(F(m[i, 0] * x + m[i, 1] * y) + m[i, 2] * z) for i in range(3))

    def apply_normal(self, n):
        im = self.invm
        x, y, z = F(n[0]), F(n[1]), F(n[2])
        return tuple(F(F(im[0, i] * x + im[1, i] * y) + im[2, i] * z) for i in range(3))


def translation(vec):
    """src/transformation.rs:215-234."""
    x, y, z = (F(c) for c in vec)
    m = np.eye(4, dtype=F)
    i = np.eye(4, dtype=F)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    i[0, 3], i[1, 3], i[2, 3] = -x, -y, -z
    return Transformation(m, i)


def scaling(vec):
    """src/transformation.rs:240-259."""
    x, y, z = (F(c) for c in vec)
    m = np.eye(4, dtype=F)
    i = np.eye(4, dtype=F)
    m[0, 0], m[1, 1], m[2, 2] = x, y, z
    i[0, 0], i[1, 1], i[2, 2] = F(F(1.0) / x), F(F(1.0) / y), F(F(1.0) / z)
    return Transformation(m, i)


def _sincos(theta):
    # f32::sin / f32::cos of an f32 angle: evaluated in binary64 and rounded once (the reference
    # calls the platform libm here; its last ulp is not pinned, SURVEY.md §8c iii).
    t = float(F(theta))
    return F(math.sin(t)), F(math.cos(t))


def rotation_x(theta):
    """src/transformation.rs:265-284 (theta in radians)."""
    s, c = _sincos(theta)
    m = np.eye(4, dtype=F)
    i = np.eye(4, dtype=F)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    i[1, 1], i[1, 2], i[2, 1], i[2, 2] = c, s, -s, c
    return Transformation(m, i)


def rotation_y(theta):
    """src/transformation.rs:290-309."""
    s, c = _sincos(theta)
    m = np.eye(4, dtype=F)
    i = np.eye(4, dtype=F)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    i[0, 0], i[0, 2], i[2, 0], i[2, 2] = c, -s, s, c
    return Transformation(m, i)


def rotation_z(theta):
    """src/transformation.rs:315-334."""
    s, c = _sincos(theta)
    m = np.eye(4, dtype=F)
    i = np.eye(4, dtype=F)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    i[0, 0], i[0, 1], i[1, 0], i[1, 1] = c, s, -s, c
    return Transformation(m, i)
