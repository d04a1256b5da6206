"""Diagnose a failing tools/fuzz_parity.py seed: which shape wins in the oracle vs on the GPU for random rays.

  python tools/fuzz_debug.py SEED [SEED ...]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib as O  # noqa: E402
import rustracer_amd as ra  # noqa: E402
from rustracer_amd import _abi  # noqa: E402
from rustracer_amd.world import flatten_world, to_rt_ray  # noqa: E402
import fuzz_parity  # noqa: E402


def main():
    for seed in map(int, sys.argv[1:]):
        world, cam, renderer, w, h, kw = fuzz_parity.random_case(seed)
        rng = np.random.default_rng(seed + 7)
        rays = []
        for _ in range(20000):
            o = rng.uniform(-6, 6, 3)
            d = rng.normal(size=3)
            rays.append(to_rt_ray(ra.Ray(origin=tuple(o), dir=tuple(d))))
        g = world.ray_intersections(rays)
        o = O.world_ray_intersection(flatten_world(world), rays, O.MATH_STRICT)
        nn, nu = C.c_uint32(), C.c_uint32()
        _abi.check(_abi.lib().rt_scene_hierarchy_info(world.device().handle, C.byref(nn), C.byref(nu)))
        bad = {}
        for a, b in zip(g, o):
            if a.hit != b.hit or (a.hit and (a.shape != b.shape or a.t != b.t)):
                key = (b.shape if b.hit else -1)
                bad.setdefault(key, []).append((a.shape if a.hit else -1, a.t, b.t))
        print(f"seed {seed}: {fuzz_parity.describe(world, cam, renderer, w, h, kw)} nodes {nn.value} unbounded {nu.value}: "
              f"{sum(len(v) for v in bad.values())} of {len(rays)} rays differ, oracle winners involved: {sorted(bad)[:10]}")
        for k in sorted(bad)[:3]:
            print("  oracle shape", k, "gpu (shape, t, oracle t):", bad[k][:3])
            if k >= 0:
                s = world.shapes[k]
                print("   kind", type(s).__name__, "m =", np.array2string(np.asarray(s.transformation.m), precision=4, max_line_width=200).replace("\n", " "))
                print("   invm =", np.array2string(np.asarray(s.transformation.invm), precision=4, max_line_width=200).replace("\n", " "))


if __name__ == "__main__":
    main()
