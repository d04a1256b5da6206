"""Host mirror of src/random.rs (PCG-XSH-RR 64/32) for host-side seeding (scene generators).

Not on the render path: the GPU kernels carry their own PCG (csrc/rt_device.h), exposed for
parity checks through rt_pcg_random.
"""
import numpy as np

_MASK64 = (1 << 64) - 1
_MULT = 6364136223846793005


class Pcg:
    """src/random.rs:7-49."""

    def __init__(self, init_state=42, init_seq=54):
        self.state = 0
        self.inc = ((init_seq << 1) | 1) & _MASK64
        self.random()
        self.state = (self.state + init_state) & _MASK64
        self.random()

    def random(self):
        old = self.state
        self.state = (old * _MULT + self.inc) & _MASK64
        xor_shifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xor_shifted >> rot) | (xor_shifted << ((-rot) & 31))) & 0xFFFFFFFF

    def random_float(self):
        return np.float32(np.float32(self.random()) / np.float32(4294967296.0))
