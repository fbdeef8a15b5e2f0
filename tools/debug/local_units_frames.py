import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from path_tracer_amd import abi
from path_tracer_amd.raytracer import Context
from oracle import binding
from tests import util as U
import tests.test_gpu_local_units as T

name = sys.argv[1] if len(sys.argv) > 1 else "scale_0.01"
floor = (abi.OBJ_SQUARE, U.translate(0, -2.5, -9) @ U.rot_x(-math.pi / 2) @ U.scale(60), U.material(kd=(0.8, 0.8, 0.8)))
all_lights = [U.dir_light((0.2, -1.0, -0.4)), U.point_light((3, 6, 2), atten=(1, 0, 0)), U.area_light((-2, 5, -6), (1, 0, 0), (0, 0, 1), 2, 2)]
gpu = Context(0)
for lights in ([], all_lights[:1], all_lights[1:2], all_lights[2:], all_lights):
    sc = T._scaled_scene(T.XFORMS[name], lights=lights, extra=[floor])
    sc.ambient = np.array([0.3, 0.3, 0.3])
    for m in sc.materials: m.ka = np.array([0.5, 0.5, 0.5])
    orc = binding.Oracle(sc)
    gpu.upload_scene(sc)
    for depth in (1, 2, 3, 4):
        p = abi.default_params(96, 96, spp=1, rr_q=0.0, max_depth=depth, flags=abi.FLAG_FIXED_AREA_PATTERN)
        f32, _ = gpu.render(p)
        ref = orc.render(p, threads=8)
        diff = np.abs(f32.astype(np.float64) - ref).max(axis=2)
        bad = np.argwhere(diff > 1e-5)
        print(len(lights), "lights depth", depth, "bad", len(bad), "max", diff.max(), "first", bad[:5].tolist(), flush=True)
