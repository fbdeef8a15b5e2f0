"""The C-ABI library loads, exports every symbol include/mipt.h declares, and the
ctypes mirror agrees with the C structs.  No compute call is made (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mipt.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mipt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("mipt_create", "mipt_upload_scene", "mipt_render", "mipt_render_accumulate", "mipt_resolve",
                 "mipt_trace_pixel", "mipt_intersect", "mipt_destroy", "mipt_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from path_tracer_amd import abi

    lib = abi.load_library()
    for name in declared_functions():
        assert name in abi.SIGNATURES, f"{name} is declared in mipt.h but has no ctypes signature"
        assert hasattr(lib, name), f"{name} is not exported by libmipt.so"
    for name in abi.SIGNATURES:
        assert name in declared_functions(), f"{name} is bound but not declared in mipt.h"


def test_struct_sizes_match_the_c_header(tmp_path):
    from path_tracer_amd import abi

    pairs = {
        "mipt_material": abi.Material, "mipt_object": abi.Object, "mipt_mesh": abi.Mesh, "mipt_light": abi.Light,
        "mipt_camera": abi.Camera, "mipt_image": abi.Image, "mipt_scene_desc": abi.SceneDesc, "mipt_params": abi.Params,
        "mipt_counters": abi.Counters, "mipt_kernel_times": abi.KernelTimes,
    }
    src = tmp_path / "sizes.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in pairs)
    offs = '  printf("off_seed %zu\\n", offsetof(mipt_params, seed));\n  printf("off_lights %zu\\n", offsetof(mipt_scene_desc, lights));'
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(void) {{\n{body}\n{offs}\n  return 0;\n}}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n, t in pairs.items():
        assert int(out[n]) == C.sizeof(t), n
    assert int(out["off_seed"]) == abi.Params.seed.offset
    assert int(out["off_lights"]) == abi.SceneDesc.lights.offset


def test_default_params_are_the_reference_literals():
    from path_tracer_amd import abi

    lib = abi.load_library()
    assert lib.mipt_version() == 1
    p = abi.Params()
    lib.mipt_default_params(C.byref(p), 512, 512)
    assert (p.width, p.height) == (512, 512)
    assert p.spp == 100            # RayTracer.cpp:61
    assert p.rr_q == 0.1           # RayTracer.cpp:229
    assert p.area_samples == 10    # light.cpp:118
    assert p.t_max == 1000.0       # scene.cpp:129
    q = abi.default_params(512, 512)
    for name, _ in abi.Params._fields_:
        assert getattr(p, name) == getattr(q, name), name


def test_render_calls_fail_loudly_without_context_or_scene():
    from path_tracer_amd import abi

    lib = abi.load_library()
    p = abi.default_params(8, 8)
    assert lib.mipt_render(None, C.byref(p), None, None) == abi.ERR_INVALID
    assert b"null context" in lib.mipt_last_error(None)
    assert lib.mipt_create(None, 0) == abi.ERR_INVALID
    h = C.c_void_p()
    rc = lib.mipt_create(C.byref(h), 0)
    if rc == 0:  # a GPU is present
        assert lib.mipt_render(h, C.byref(p), None, None) == abi.ERR_NOSCENE
        lib.mipt_destroy(h)
    else:
        assert rc == abi.ERR_DEVICE and lib.mipt_last_error(None)


def test_product_package_never_touches_the_oracle():
    """The render path must not route through the CPU checker."""
    pkg = os.path.join(ROOT, "path_tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f"{f} mentions the oracle"
    host = os.path.join(ROOT, "host")
    if os.path.isdir(host):
        for f in os.listdir(host):
            if f.endswith((".cpp", ".h")):
                assert "oracle" not in open(os.path.join(host, f)).read(), f


# ---- libmiptdist.so (include/mipt_dist.h): the multi-GPU render over RCCL ----------------------------------
DIST_HEADER = os.path.join(ROOT, "include", "mipt_dist.h")


def load_dist():
    from path_tracer_amd import abi

    abi.load_library()  # libmiptdist.so resolves libmipt.so by soname
    return C.CDLL(os.path.join(ROOT, "path_tracer_amd", "libmiptdist.so"))


def test_dist_library_exports_every_declared_symbol():
    text = re.sub(r"/\*.*?\*/", "", open(DIST_HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mipt_dist_[a-z_0-9]+)\s*\(", text)))
    assert {"mipt_dist_new_id", "mipt_dist_init", "mipt_dist_render", "mipt_dist_shard_params", "mipt_dist_destroy"} <= set(names)
    lib = load_dist()
    for name in names:
        assert hasattr(lib, name), f"{name} is not exported by libmiptdist.so"


def test_dist_shard_plans_partition_the_job():
    """mipt_dist_shard_params (host arithmetic, no device): the ranks' shares are disjoint and cover the job, and they
    are the shares path_tracer_amd.dist.shard_params gives the Python ranks."""
    from path_tracer_amd import abi
    from path_tracer_amd.dist import shard_params

    lib = load_dist()
    lib.mipt_dist_shard_params.restype = C.c_int
    lib.mipt_dist_shard_params.argtypes = [C.POINTER(abi.Params), C.c_int, C.c_int, C.c_int, C.POINTER(abi.Params)]
    base = abi.default_params(64, 37, spp=10, band_rows=4)
    base.sample_begin, base.sample_end = 2, 9
    for world in (1, 2, 3, 8):
        for mode, name in ((0, "samples"), (1, "tiles")):
            for r in range(world):
                mine = abi.Params()
                assert lib.mipt_dist_shard_params(C.byref(base), r, world, mode, C.byref(mine)) == 0
                want = shard_params(base, r, world, name)
                for field, _ in abi.Params._fields_:
                    assert getattr(mine, field) == getattr(want, field), (world, name, r, field)
    mine = abi.Params()
    assert lib.mipt_dist_shard_params(C.byref(base), 2, 2, 0, C.byref(mine)) == abi.ERR_INVALID
    sharded = abi.default_params(8, 8, shard_index=1, shard_count=2)
    assert lib.mipt_dist_shard_params(C.byref(sharded), 0, 2, 1, C.byref(mine)) == abi.ERR_INVALID
