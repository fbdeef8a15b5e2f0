"""mipt_render_accumulate_static: the render entry point that allocates nothing (its workspace is a
static device array of the library).  Through the C ABI, against mipt_render_accumulate on the same
inputs: the same samples from the same random streams, so the per-pixel sums agree to the rounding
of fp64 additions made in a different order (<= 1e-12 on sums of values in [0, 1]).  Two calls in a
row, the second on a smaller frame, so that what the first left in the array cannot leak into the
second; then the limits it must refuse.
"""
import numpy as np
import pytest

from path_tracer_amd import abi
from tests import util as U

pytestmark = pytest.mark.gpu


def _both(gpu, params):
    import torch

    n = 3 * params.width * params.height
    a = torch.zeros(n, dtype=torch.float64, device="cuda")
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    gpu.render_accumulate_static(params, a.data_ptr())
    gpu.render_accumulate(params, b.data_ptr())
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_workspace_render_matches_the_allocating_one(gpu, scenes):
    import torch

    gpu.upload_scene(scenes("spheres"))                               # area light: 10 shadow rays per hit
    big = abi.default_params(200, 160, spp=6, seed=3)                  # 192 000 samples: twelve batches of the static workspace, some of which spill
    a, b = _both(gpu, big)
    assert b.mean() > 0.05
    assert np.abs(a - b).max() <= 1e-12
    small = abi.default_params(48, 40, spp=3, seed=4, area_samples=4)  # smaller frame, other light table
    a, b = _both(gpu, small)
    assert b.mean() > 0.05
    assert np.abs(a - b).max() <= 1e-12
    # no device allocation: the card's free memory does not move over static calls (the allocating
    # calls above have already made the runtime set up whatever the kernels themselves need)
    d = torch.zeros(3 * 200 * 160, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    gpu.render_accumulate_static(big, d.data_ptr())
    gpu.render_accumulate_static(small, d.data_ptr())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    # shards and sample ranges add up like the allocating entry point's
    full = torch.zeros(3 * 48 * 40, dtype=torch.float64, device="cuda")
    for r in range(2):
        for (s0, s1) in ((0, 1), (1, 3)):
            gpu.render_accumulate_static(abi.default_params(48, 40, spp=3, seed=4, area_samples=4, shard_index=r, shard_count=2, band_rows=4,
                                                            sample_begin=s0, sample_end=s1), full.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(full.cpu().numpy() - b).max() <= 1e-12


def test_static_workspace_on_meshes_and_translucent_scenes(gpu):
    from path_tracer_amd import synth

    gpu.upload_scene(synth.make_scene(20000, seed=7))
    a, b = _both(gpu, abi.default_params(128, 72, spp=5, seed=2))
    assert b.mean() > 0.02 and np.abs(a - b).max() <= 1e-12
    glass = U.material(kd=(0.2, 0.2, 0.2), kt=(0.5, 0.25, 0.9))
    sc = U.scene([(abi.OBJ_BOX, U.translate(0, 2, 0) @ U.scale(3, 1, 3), glass), (abi.OBJ_SQUARE, U.rot_x(-np.pi / 2) @ U.scale(30), U.material())],
                 lights=[U.point_light((0.5, 9, 0.5), atten=(1, 0, 0)), U.area_light((-1, 8, -1), (1, 0, 0), (0, 0, 1), 2, 2)],
                 cam=U.camera(eye=(0, 6, 12), viewdir=(0, -0.4, -1)))
    gpu.upload_scene(sc)
    a, b = _both(gpu, abi.default_params(64, 64, spp=3))
    assert b.mean() > 0.02 and np.abs(a - b).max() <= 1e-12


def test_static_workspace_refuses_what_it_cannot_hold(gpu, scenes):
    import torch
    from path_tracer_amd.raytracer import MiptError

    gpu.upload_scene(scenes("spheres"))
    d = torch.zeros(3 * 8 * 8, dtype=torch.float64, device="cuda")
    for bad, code in ((dict(flags=abi.FLAG_COUNTERS), abi.ERR_INVALID), (dict(area_samples=4096 * 17), abi.ERR_INVALID)):
        with pytest.raises(MiptError) as e:
            gpu.render_accumulate_static(abi.default_params(8, 8, **bad), d.data_ptr())
        assert e.value.code == code
    with pytest.raises(MiptError) as e:                                # 16385 rows
        gpu.render_accumulate_static(abi.default_params(1, 16385, spp=1), 1)
    assert e.value.code == abi.ERR_LIMIT and "rows" in str(e.value)
    # two facing mirrors: the tree doubles per level.  The deepest rays are traced first, so a batch of 2^14 samples parks at most
    # 2^14 rays per level: the fixed overflow stack (2^18) holds any tree of depth <= 16 and is outgrown here at level 17;
    # the allocating call grows its own stack, this one says so
    mirror = U.material(kd=(0.3, 0.3, 0.3), ka=(0.1, 0.1, 0.1), roughness=0.0)
    sc = U.scene([(abi.OBJ_SQUARE, U.translate(0, 0, -2) @ U.scale(400), mirror), (abi.OBJ_SQUARE, U.translate(0, 0, 2) @ U.scale(400), mirror)],
                 lights=[U.point_light((0, 0.5, 0), atten=(1, 0, 0))], cam=U.camera(eye=(0, 0, 0)), ambient=(1, 1, 1))
    gpu.upload_scene(sc)
    d = torch.zeros(3 * 64 * 64, dtype=torch.float64, device="cuda")
    with pytest.raises(MiptError) as e:
        gpu.render_accumulate_static(abi.default_params(64, 64, spp=8, rr_q=0.0, max_depth=48), d.data_ptr())
    assert e.value.code == abi.ERR_LIMIT and "static workspace" in str(e.value)
    gpu.upload_scene(scenes("spheres"))                                # and the context still works afterwards
    a, b = _both(gpu, abi.default_params(32, 32, spp=2))
    assert np.abs(a - b).max() <= 1e-12
