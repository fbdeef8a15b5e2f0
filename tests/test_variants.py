"""The two authored variants of the demo scenes: cornellBoxes under a rectangular area light
(10 shadow rays per hit) and hitchcockBRDF with its boxes as triangle meshes.  CPU checks
with the oracle; the GPU side of the same scenes is in test_gpu_parity.py."""
import numpy as np

from path_tracer_amd import abi, synth
from tests import util as U


def test_tessellated_boxes_are_the_same_surface(scenes, oracle_lib):
    for name in ("cornellBoxes", "hitchcockBRDF"):
        sc = scenes(name)
        tv = synth.tessellate_boxes(sc)
        n_boxes = sum(o.type == abi.OBJ_BOX for o in sc.objects)
        assert n_boxes > 0 and tv.n_triangles() == 12 * n_boxes
        assert not any(o.type == abi.OBJ_BOX for o in tv.objects)
        a, b = oracle_lib.Oracle(sc), oracle_lib.Oracle(tv)
        org, d = U.camera_rays(sc, 5000, seed=3)
        ta, pa, na = a.intersect(org, d)
        tb, pb, nb = b.intersect(org, d)
        assert np.array_equal(pa >= 0, pb >= 0)
        hit = pa >= 0
        assert np.abs(ta[hit] - tb[hit]).max() < 1e-9
        assert np.abs(na[hit] - nb[hit]).max() < 1e-9


def test_area_light_variant_lights_the_cornell_box(scenes, oracle_lib):
    sc = scenes("cornellBoxes")
    av = synth.cornell_area_light(sc)
    assert [L.type for L in av.lights] == [abi.LIGHT_RECT_AREA]
    assert [L.type for L in sc.lights] == [abi.LIGHT_POINT]  # the input is untouched
    p = abi.default_params(48, 48, spp=4)
    img, cnt = oracle_lib.Oracle(av).render(p, threads=8, want_counters=True)
    ref, cnt0 = oracle_lib.Oracle(sc).render(p, threads=8, want_counters=True)
    assert img.mean() > 0.05
    # ten shadow rays per shaded hit instead of one
    assert cnt["shadow_rays"] == 10 * cnt["shaded_hits"]
    assert cnt0["shadow_rays"] == cnt0["shaded_hits"]
