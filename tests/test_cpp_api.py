"""The reference's RenderingTest.* suite compiled against the C++ mirror of the reference API
(tests/cpp/rendering_tests.cpp).  CPU: must compile and link; GPU: must pass."""
import os
import subprocess

import numpy as np
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


DEMO = os.path.join(ROOT, "raytracer_amd", "lib", "rt_demo")


def test_headless_demo_fails_loudly_without_a_gpu(built):
    """The headless Demo (host/Demo/headless/Main.cpp: LoadScene -> Viewport -> GetFrontBuffer -> BMP) is built by build();
    on a machine without a GPU it must say so instead of rendering on some fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    scene = os.path.join(ROOT, "tests", "golden", "obj", "scene.json")
    r = subprocess.run([DEMO, "-s", scene, "--data", os.path.dirname(scene) + "/", "-w", "64", "-h", "48", "--passes", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "not available" in r.stderr


@pytest.mark.gpu
def test_headless_demo_renders_the_ingested_scene(built, tmp_path):
    scene = os.path.join(ROOT, "tests", "golden", "obj", "scene.json")
    out = tmp_path / "out.bmp"
    r = subprocess.run([DEMO, "-s", scene, "--data", os.path.dirname(scene) + "/", "-w", "160", "-h", "100", "--passes", "8", "--depth", "5", "--output", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Msamples/s" in r.stdout
    data = out.read_bytes()
    assert data[:2] == b"BM" and len(data) == 54 + 160 * 3 * 100
    pixels = np.frombuffer(data[54:], dtype=np.uint8)
    assert pixels.max() > 100 and len(np.unique(pixels)) > 50      # an image, not a constant


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", ["Path Tracer", "Light Tracer", "VCM", "Debug"])
def test_headless_demo_other_renderers(built, tmp_path, renderer):
    """The reference Demo's --renderer option with every other name of CreateRenderer (Renderer.cpp:45-69)."""
    scene = os.path.join(ROOT, "tests", "golden", "obj", "scene.json")
    out = tmp_path / "out.bmp"
    r = subprocess.run([DEMO, "-s", scene, "--data", os.path.dirname(scene) + "/", "-w", "128", "-h", "80", "--passes", "6", "--depth", "5", "--renderer", renderer,
                        "--output", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    data = out.read_bytes()
    assert data[:2] == b"BM" and len(data) == 54 + 128 * 3 * 80
    pixels = np.frombuffer(data[54:], dtype=np.uint8)
    assert pixels.max() > 60 and len(np.unique(pixels)) > 20
