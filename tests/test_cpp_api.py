"""The reference's RenderingTest.* suite compiled against the C++ mirror of the reference API
(tests/cpp/rendering_tests.cpp).  CPU: must compile and link; GPU: must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "rendering_tests")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib = os.path.join(ROOT, "raytracer_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "raytracer_amd", "host"),
                           os.path.join(ROOT, "tests", "cpp", "rendering_tests.cpp"), "-o", EXE, "-L" + lib,
                           "-lraytracer_amd_host", "-lrtgpu", "-Wl,-rpath," + lib])


def test_reference_style_cpp_tests_compile_and_link(built):
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_reference_rendering_tests_pass_on_gpu(built):
    _build()
    env = dict(os.environ, RT_DATA_DIR=os.path.join(ROOT, "raytracer_amd", "data"), RT_SEED="12345")
    out = subprocess.run([EXE], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 test(s) failed" in out.stdout
