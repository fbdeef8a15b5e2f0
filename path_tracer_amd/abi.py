"""ctypes mirror of include/mipt.h (POD structs and the libmipt.so entry points).

Every structure here must stay field-for-field identical to the C header; the
CPU test `tests/test_abi.py` checks sizes against the compiled library.
"""
from __future__ import annotations

import ctypes as C
import os

OBJ_SPHERE, OBJ_BOX, OBJ_SQUARE, OBJ_TRIMESH = 0, 1, 2, 3
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_RECT_AREA = 0, 1, 2
FLAG_FIXED_AREA_PATTERN = 1
FLAG_COUNTERS = 2
FLAG_PROFILE = 4
FLAG_CULL_NEGLIGIBLE = 8
K_GENERATE, K_INTERSECT, K_SHADE, K_SHADOW, K_RESOLVE, K_SHADOW_SUM, K_TAIL, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
KERNEL_NAMES = ("generate", "intersect", "shade", "shadow", "resolve", "shadow_sum", "tail")

OK, ERR_INVALID, ERR_NOSCENE, ERR_DEVICE, ERR_NOMEM, ERR_LIMIT = 0, -1, -2, -3, -4, -5

c_double3 = C.c_double * 3
c_float3 = C.c_float * 3


class Material(C.Structure):
    _fields_ = [
        ("kd", c_double3),
        ("ka", c_double3),
        ("kt", c_double3),
        ("roughness", C.c_double),
        ("metallic", C.c_double),
        ("index", C.c_double),
        ("kd_texture", C.c_int32),
        ("ka_texture", C.c_int32),
    ]


class Object(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("material", C.c_int32),
        ("mesh", C.c_int32),
        ("reserved", C.c_int32),
        ("xform", C.c_double * 16),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(C.c_double)),
        ("n_vertices", C.c_int64),
        ("faces", C.POINTER(C.c_int32)),
        ("n_faces", C.c_int64),
        ("normals", C.POINTER(C.c_double)),
        ("uvs", C.POINTER(C.c_double)),
        ("colors", C.POINTER(C.c_double)),
    ]


class Light(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("reserved", C.c_int32),
        ("color", c_double3),
        ("position", c_double3),
        ("direction", c_double3),
        ("u", c_double3),
        ("v", c_double3),
        ("uL", C.c_double),
        ("vL", C.c_double),
        ("atten", c_float3),
        ("pad", C.c_float),
    ]


class Camera(C.Structure):
    _fields_ = [("eye", c_double3), ("look", c_double3), ("u", c_double3), ("v", c_double3)]


class Image(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("ambient", c_double3),
        ("materials", C.POINTER(Material)),
        ("n_materials", C.c_int32),
        ("objects", C.POINTER(Object)),
        ("n_objects", C.c_int32),
        ("meshes", C.POINTER(Mesh)),
        ("n_meshes", C.c_int32),
        ("lights", C.POINTER(Light)),
        ("n_lights", C.c_int32),
        ("cubemap", C.POINTER(Image)),
        ("textures", C.POINTER(Image)),
        ("n_textures", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("spp", C.c_int32),
        ("area_samples", C.c_int32),
        ("rr_q", C.c_double),
        ("t_max", C.c_double),
        ("max_depth", C.c_int32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("shard_index", C.c_int32),
        ("shard_count", C.c_int32),
        ("band_rows", C.c_int32),
        ("sample_begin", C.c_int32),
        ("sample_end", C.c_int32),
        ("batch_samples", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("path_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("shadow_node_visits", C.c_uint64),
        ("shadow_prim_tests", C.c_uint64),
        ("ray_stages", C.c_uint64),
        ("nan_samples", C.c_uint64),
    ]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * K_COUNT), ("launches", C.c_uint64 * K_COUNT), ("items", C.c_uint64 * K_COUNT)]

    def as_dict(self):
        return {KERNEL_NAMES[k]: {"ms": float(self.ms[k]), "launches": int(self.launches[k]), "items": int(self.items[k])} for k in range(K_COUNT)}


def default_params(width: int, height: int, **over) -> Params:
    """The reference's literals: spp 100 (RayTracer.cpp:61), q 0.1 (:229), 10 area
    samples (light.cpp:118), t_max 1000 (scene.cpp:129); one shard, all samples."""
    p = Params()
    p.width, p.height = width, height
    p.spp = 100
    p.area_samples = 10
    p.rr_q = 0.1
    p.t_max = 1000.0
    p.max_depth = 64
    p.flags = 0
    p.seed = 1
    p.shard_index, p.shard_count, p.band_rows = 0, 1, 8
    p.sample_begin, p.sample_end = 0, p.spp
    p.batch_samples = 0
    explicit_range = "sample_begin" in over or "sample_end" in over
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    if not explicit_range:
        p.sample_begin, p.sample_end = 0, p.spp
    return p


LIB_NAME = "libmipt.so"


def lib_path() -> str:
    """libmipt.so next to this file; MIPT_LIB names another build of the same library
    (kernel tuning experiments)."""
    return os.environ.get("MIPT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


# name -> (restype, argtypes); every symbol include/mipt.h declares.
SIGNATURES = {
    "mipt_version": (C.c_int, []),
    "mipt_last_error": (C.c_char_p, [C.c_void_p]),
    "mipt_default_params": (None, [C.POINTER(Params), C.c_int, C.c_int]),
    "mipt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "mipt_destroy": (None, [C.c_void_p]),
    "mipt_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mipt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "mipt_render": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]),
    "mipt_render_accumulate": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]),
    "mipt_render_accumulate_static": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]),
    "mipt_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mipt_trace_pixel": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "mipt_hit_material": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mipt_intersect": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mipt_shadow_attenuation": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mipt_shadow_rays": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mipt_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "mipt_reset_counters": (C.c_int, [C.c_void_p]),
    "mipt_get_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(KernelTimes)]),
    "mipt_scene_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None


def load_library():
    """Load libmipt.so from the package directory.  There is no fallback: if the HIP
    library has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)
    this raises."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP library first (make -C path_tracer_amd/csrc). "
            "There is no CPU fallback for the render path."
        )
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and finds no GPU
    # if another copy initialised the device first, so let torch load its runtime before
    # libmipt.so resolves the same soname (the host layer uses torch for device buffers,
    # streams and torch.distributed anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
