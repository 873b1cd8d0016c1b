"""The DEVICE against what the reference's own renderer produced (tests/golden/ref_render/*.bin, see test_reference_images.py): the
rt::Viewport mirror drives the HIP path tracer with the same seed through the same call sequence; tolerance as stated there (the
reference's _mm_rcp_ss / _mm_rsqrt_ps sites are exact operations on the device)."""
import numpy as np
import pytest

import ref_scenes
import raytracer_amd as ra
from test_reference_images import COUNTER_TOL, FLOORS, compare_with_reference, load_fixture

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
