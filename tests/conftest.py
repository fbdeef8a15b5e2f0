import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the library reads its tuning and test knobs (MIPT_LDS_STACK, MIPT_BUILDER, ...) only in a process that says so
os.environ.setdefault("MIPT_DEBUG", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_NAMES = ("spheres", "spheresDirectional", "cornellBoxes", "hitchcockBRDF", "singleSphere", "areaLight", "room", "russianRouletteTest")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_scene(name):
    from path_tracer_amd.scene import Scene

    return Scene.load_npz(os.path.join(GOLDEN, "scenes", name + ".npz"))


@pytest.fixture(scope="session")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_scene(name)
        return cache[name]

    return get


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding

    binding.build()
    return binding


@pytest.fixture(scope="session")
def oracles(scenes, oracle_lib):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = oracle_lib.Oracle(scenes(name))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def gpu():
    """A rendering context on cuda:0 through the C ABI.  No fallback: if libmipt.so is
    missing or no GPU is visible the GPU tests fail."""
    from path_tracer_amd.raytracer import Context

    return Context(0)
