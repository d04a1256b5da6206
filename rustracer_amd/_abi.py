"""ctypes view of include/rustracer_amd.h and the loader of librustracer_amd.so.

The product path has no CPU fallback: if the HIP library is missing or fails to load,
`lib()` raises (it never routes to the oracle).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RUSTRACER_AMD_SO: diagnostics only (an experimental build of the same library)
LIB_PATH = os.environ.get("RUSTRACER_AMD_SO") or os.path.join(_HERE, "librustracer_amd.so")

# enums (include/rustracer_amd.h)
RT_SHAPE_SPHERE, RT_SHAPE_PLANE = 0, 1
RT_PIGMENT_UNIFORM, RT_PIGMENT_CHECKERED, RT_PIGMENT_IMAGE = 0, 1, 2
RT_BRDF_DIFFUSE, RT_BRDF_SPECULAR = 0, 1
RT_CAMERA_ORTHOGONAL, RT_CAMERA_PERSPECTIVE = 0, 1
RT_ALGO_ONOFF, RT_ALGO_FLAT, RT_ALGO_PATHTRACER, RT_ALGO_DUMMY = 0, 1, 2, 3
RT_NOALLOC_SCRATCH_BYTES = 8 * 1024 * 1024 + 4096
RT_NOALLOC_MAX_DEPTH = 513


class rt_shape(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32),
                ("m", C.c_float * 16), ("invm", C.c_float * 16)]


class rt_pigment(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("c1", C.c_float * 3), ("c2", C.c_float * 3),
                ("steps", C.c_uint32), ("texture", C.c_uint32)]


class rt_material(C.Structure):
    _fields_ = [("brdf", C.c_uint32), ("brdf_pigment", rt_pigment), ("emitted", rt_pigment)]


class rt_texture(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class rt_camera(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("distance", C.c_float), ("aspect", C.c_float),
                ("m", C.c_float * 16)]


class rt_renderer(C.Structure):
    _fields_ = [("algorithm", C.c_uint32), ("bg", C.c_float * 3), ("fg", C.c_float * 3),
                ("num_of_rays", C.c_uint32), ("max_depth", C.c_uint32), ("rr_limit", C.c_uint32)]


class rt_ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3),
                ("tmin", C.c_float), ("tmax", C.c_float), ("depth", C.c_uint32)]


class rt_hit(C.Structure):
    _fields_ = [("hit", C.c_uint32), ("shape", C.c_uint32), ("world_point", C.c_float * 3),
                ("normal", C.c_float * 3), ("uv", C.c_float * 2), ("t", C.c_float)]


class rt_pcg(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


class rt_partition(C.Structure):
    _fields_ = [("part", C.c_uint32), ("n_parts", C.c_uint32), ("tile_rows", C.c_uint32), ("rotate", C.c_uint32),
                ("in_place", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64),
                ("diffuse_scatters", C.c_uint64), ("specular_scatters", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("rays_shared", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


P = C.POINTER
# name -> (restype, argtypes); exactly the symbols include/rustracer_amd.h declares
SIGNATURES = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_scene_create": (C.c_void_p, [P(rt_shape), C.c_uint32, P(rt_material), C.c_uint32,
                                     P(rt_texture), C.c_uint32]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_set_schedule": (C.c_int, [C.c_void_p, P(C.c_uint32), C.c_uint32]),
    "rt_partition_pixels": (C.c_uint64, [C.c_uint32, C.c_uint32, P(rt_partition)]),
    "rt_scene_hierarchy_info": (C.c_int, [C.c_void_p, P(C.c_uint32), P(C.c_uint32)]),
    "rt_render": (C.c_int, [C.c_void_p, P(rt_camera), P(rt_renderer), C.c_uint32, C.c_uint32,
                            C.c_uint64, C.c_uint64, C.c_uint32, P(rt_partition), C.c_void_p,
                            C.c_void_p, P(rt_stats)]),
    "rt_render_frames": (C.c_int, [C.c_void_p, P(rt_camera), C.c_uint32, P(rt_renderer), C.c_uint32, C.c_uint32,
                                   C.c_uint64, C.c_uint64, C.c_uint32, P(rt_partition), P(C.c_void_p),
                                   C.c_void_p, P(rt_stats)]),
    "rt_render_noalloc": (C.c_int, [C.c_void_p, P(rt_camera), P(rt_renderer), C.c_uint32, C.c_uint32,
                                    C.c_uint64, C.c_uint64, C.c_uint32, P(rt_partition), C.c_void_p,
                                    C.c_void_p]),
    "rt_unpack_parts": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_fire_all_rays": (C.c_int, [P(rt_shape), C.c_uint32, P(rt_material), C.c_uint32,
                                   P(rt_texture), C.c_uint32, P(rt_camera), P(rt_renderer),
                                   C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                   P(C.c_float), P(rt_stats)]),
    "rt_fire_release": (None, []),
    "rt_last_fire_devices": (C.c_uint32, [P(C.c_int32), P(C.c_double), C.c_uint32]),
    "rt_world_ray_intersection": (C.c_int, [C.c_void_p, P(rt_ray), C.c_uint64, P(rt_hit)]),
    "rt_renderer_solve": (C.c_int, [C.c_void_p, P(rt_renderer), P(rt_ray), P(rt_pcg), C.c_uint64,
                                    P(C.c_float)]),
    "rt_brdf_scatter_ray": (C.c_int, [C.c_uint32, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float), C.c_uint32,
                                      C.c_uint64, P(rt_ray)]),
    "rt_pcg_random": (C.c_int, [P(C.c_uint64), P(C.c_uint64), C.c_uint64, C.c_uint32,
                                P(C.c_uint32)]),
    "rt_strict_math_eval": (C.c_int, [C.c_int, P(C.c_float), P(C.c_float), C.c_uint64,
                                      P(C.c_float)]),
}

_lib = None


def lib():
    """Load librustracer_amd.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; two HSA runtimes in one
        # process cannot both open the GPU.  Import torch first so that librustracer_amd.so binds
        # to the runtime already in the process (torch is the plumbing for device memory, streams
        # and torch.distributed; a C/C++ host without torch simply gets /opt/rocm's runtime).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class RtError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise RtError(lib().rt_last_error().decode("utf-8", "replace"))
