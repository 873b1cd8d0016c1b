"""The DEVICE against what the reference's own renderer produced (tests/golden/ref_render/*.bin, see test_reference_images.py): the
rt::Viewport mirror drives the HIP path tracer with the same seed through the same call sequence; tolerance as stated there (the
reference's _mm_rcp_ss / _mm_rsqrt_ps sites are exact operations on the device)."""
import numpy as np
import pytest

import ref_scenes
import raytracer_amd as ra
from test_reference_images import COUNTER_TOL, FLOORS, compare_statistically, compare_with_reference, load_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(ref_scenes.FIXTURES))
def test_device_image_matches_the_reference_renderer(built, name, walk):
    """`walk`: "default" = the library as shipped and as bench.py times it (intersection counters off like the reference's default build,
    Core/Config.h:4: k_trace_wide + dense path state on `mesh_single`); "counting" = the reference's binary walk with its
    RT_ENABLE_INTERSECTION_COUNTERS counters."""
    fx = load_fixture(name)
    scene, camera = ref_scenes.FIXTURES[name][0](fx["w"] / fx["h"])
    vp = ra.Viewport(fx["w"], fx["h"], seed=ref_scenes.SEED, max_ray_depth=fx["depth"], dimensions=fx["dims"], light_sampling_all=fx["sampling_all"])
    vp.set_renderer(scene, intersection_counters=(walk == "counting"))     # CreateRenderer + SetRenderer + Reset, like the reference's callers
    vp.render(camera, fx["passes"])
    img = vp.sum_buffer()
    compare_with_reference(fx, img, vp.counters(), FLOORS[name], COUNTER_TOL.get(name, 0.001))


@pytest.mark.parametrize("name", sorted(ref_scenes.STATISTICAL_FIXTURES))
def test_device_matches_the_reference_renderer_statistically_under_single_with_two_lights(built, name, walk):
    """BASELINE config 3's light-picking mode -- `Single`, background + directional light -- against the reference's own one-thread frame
    (tests/golden/ref_render/mesh_2k_single.bin, 4096 passes): tolerance in test_reference_images.compare_statistically."""
    fx = load_fixture(name)
    scene, camera = ref_scenes.STATISTICAL_FIXTURES[name][0](fx["w"] / fx["h"])
    vp = ra.Viewport(fx["w"], fx["h"], seed=ref_scenes.SEED, max_ray_depth=fx["depth"], dimensions=fx["dims"], light_sampling_all=fx["sampling_all"])
    vp.set_renderer(scene, intersection_counters=(walk == "counting"))
    vp.render(camera, fx["passes"])
    compare_statistically(fx, vp.sum_buffer(), vp.counters())
