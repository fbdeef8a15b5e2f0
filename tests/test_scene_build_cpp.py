"""Host-side check of the scene builder: compiles tests/cpp/check_bvh.cpp against
path_tracer_amd/csrc/scene_build.cpp with g++ and runs it (no GPU involved)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quantised_bvh4_is_complete_and_conservative(tmp_path):
    exe = str(tmp_path / "check_bvh")
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "check_bvh.cpp"),
           os.path.join(ROOT, "path_tracer_amd", "csrc", "scene_build.cpp"), "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")
