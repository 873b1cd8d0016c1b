"""GPU parity: the HIP wavefront path tracer (through the C-ABI, via the rt::Viewport mirror) against the CPU
oracle on identical scenes and identical per-pass constants.

Bar: BIT-EXACT float3 sum buffers and IDENTICAL ray counters.  Both sides evaluate the reference's arithmetic in
the reference's operation order with IEEE ops (the two x86 approximate instructions the reference uses,
_mm_rcp_ss and _mm_rsqrt_ps, are exact ops on both sides), so there is no tolerance to state.  Tolerance versus
the REFERENCE itself: see tests/test_oracle_kat.py (2^-11 relative on the rsqrt-affected frames only)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import scene_zoo
import raytracer_amd as ra
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

COMPARED = ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays", "numRayBoxTests", "numPassedRayBoxTests",
            "numRayTriangleTests", "numPassedRayTriangleTests", "numMeshHits", "numAnalyticHits", "numShadowRayBoxTests",
            "numShadowRayTriangleTests")


NOT_INTERSECTION = ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays", "numMeshHits", "numAnalyticHits")


def run_both(scene, camera, w, h, passes, seed=99, threads=8, walk="counting", **vp_args):
    """`walk` (tests/conftest.py): "default" = the library as shipped and as bench.py times it -- intersection counters off (the
    reference's default build, Core/Config.h:4), i.e. the 4-wide walks with their re-trace hand-over and dense path state; "counting" =
    the reference's binary walk with the box / triangle test counters on."""
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=seed, **vp_args)
    vp.set_renderer(scene, intersection_counters=(walk == "counting"))
    ref = np.zeros((h, w, 3), dtype=np.float32)
    ref2 = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    for _ in range(passes):
        p = vp.next_pass_params(camera)
        vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=threads)
    img, img2 = vp.sum_buffer(secondary=True)
    return img, img2, vp.counters(), ref, ref2, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES)}, walk


def assert_identical(img, img2, counters, ref, ref2, ref_counters, walk="counting"):
    assert np.isfinite(ref).all()
    nbad = int(np.count_nonzero(img.view(np.uint32) != ref.view(np.uint32)))
    assert nbad == 0, "%d of %d sum-buffer values differ (max abs %.3e)" % (nbad, ref.size, float(np.abs(img - ref).max()))
    assert np.array_equal(img2.view(np.uint32), ref2.view(np.uint32))
    for n in (COMPARED if walk == "counting" else NOT_INTERSECTION):
        assert counters[n] == ref_counters[n], (n, counters[n], ref_counters[n])
    if walk != "counting":   # the reference's intersection counters belong to its own walk: off means untouched
        assert counters["numRayBoxTests"] == 0 and counters["numRayTriangleTests"] == 0


def test_cornell_box_bit_exact(built, walk):
    """BASELINE config 1 geometry (10 instances, top-level BVH, glass/metal/diffuse, rect light), depth 4."""
    w, h = 160, 120
    scene, camera = scenes.cornell_box(w / h)
    out = run_both(scene, camera, w, h, walk=walk, passes=4, max_ray_depth=4)
    assert_identical(*out)
    assert out[2]["numRays"] > 4 * w * h * 2


def test_sphere_area_light_bit_exact(built, walk):
    """BASELINE config 2 (single sphere + area light: no BVH, fp64 sphere intersection)."""
    w, h = 192, 108
    scene, camera = scenes.sphere_area_light(w / h)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=4, max_ray_depth=4))


def test_every_light_and_bsdf_all_strategy(built, walk):
    """All five light types, all nine BSDFs, LightSamplingStrategy::All with dimensions raised to 128."""
    w, h = 128, 96
    scene, camera = scene_zoo.all_lights_scene(w / h)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=6, dimensions=128, light_sampling_all=True))


def test_single_strategy_many_lights_and_dimension_overflow(built, walk):
    """Single strategy with 8 lights (per-pixel fallback generator picks the light) and only 16 Halton
    dimensions (samples past them come from the fallback generator, GenericSampler.cpp:106-109)."""
    w, h = 96, 64
    scene, camera = scene_zoo.all_lights_scene(w / h)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=5, dimensions=16, use_blue_noise=False))


def _mesh_under_a_delta_sun(aspect, orientation, triangles=20000, angle_degrees=0.0):
    """The Sponza-class mesh under ONE directional light (delta for angle 0) with the given Euler orientation, plus the background light."""
    pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(triangles, 7, refine=True)
    scene = ra.Scene()
    mats = [scene.add_material("diffuse", c) for _, c in scenes.SPONZA_MATERIALS]
    scene.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
    scene.add_background_light((0.3, 0.4, 0.5))
    scene.add_directional_light((8.0, 7.5, 7.0), np.float32(angle_degrees) / np.float32(180.0) * np.float32(3.14159265359), ra.transform_from_euler((0.0, 0.0, 0.0), orientation))
    scene.build()
    return scene, ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), aspect, 65.0)


@pytest.mark.parametrize("orientation", [(0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (0.0, 90.0, 0.0), (45.0, 0.0, 0.0), (80.0, 0.0, 0.0)])
def test_axis_parallel_next_event_rays(built, walk, orientation):
    """Round 5.  A sun that shines exactly along z (identity orientation): EVERY next-event ray towards it has two direction components of exactly zero; pitched or turned by 90 degrees the Euler matrix leaves one exact zero and one of 1e-7; a sun in a
    coordinate plane (pitch 45 / 80 degrees: sponza.json's orientation) gives one.  The reference's slab test drops such an axis (inf - inf) and its walk
    visits the scene's whole column there -- or, depending on the signs of plane and origin, rejects every box that reaches across zero (operand order of
    _mm_min_ps / _mm_max_ps: a ray with a negative coordinate on the dropped axis passes through most of the scene).  With the counters off such rays go from the
    4-wide walk to the re-trace launch, whose binary walk skips boxes clearly off the ray's fixed coordinate (boxNearDegenerateAxes); with the counters on the
    reference's walk runs untouched.  Images and ray / shadow-ray / hit counters are the oracle's either
    way, and the delta sun must actually light the frame (rays that are dropped or always occluded would pass a comparison of two black images)."""
    w, h = 160, 96
    scene, camera = _mesh_under_a_delta_sun(w / h, orientation)
    out = run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=5)
    assert_identical(*out)
    assert out[2]["numShadowRays"] > out[2]["numShadowRaysHit"] > 0
    # the same with a one-degree sun (cone samples around the axis: a few rays with an exactly-zero component among ordinary ones) under `All`
    scene, camera = _mesh_under_a_delta_sun(w / h, orientation, angle_degrees=1.0)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=4, light_sampling_all=True, dimensions=64))


def _sliver_mesh_under_a_delta_sun(aspect, orientation, num_slivers=6000, seed=11):
    """A floor and two walls under a delta sun, filled with near-degenerate SLIVER triangles: one long edge almost along a coordinate axis (1 to 6 units, bent off it
    by 1e-6 ... 1e-3), a third vertex 1e-5 ... 1e-2 away from the first -- extents on the short axes down to the grid's resolution, many of them lying within
    rounding distance of the fixed coordinate of an axis-parallel next-event ray (the hit points they are cast from lie ON such slivers)."""
    rng = np.random.RandomState(seed)
    mb = scenes.MeshBuilder()
    mb.add_grid((-6.0, 0.0, 6.0), (12.0, 0, 0), (0, 0, -12.0), 6, 6, 0, 4.0)        # floor
    mb.add_grid((-6.0, 0.0, -6.0), (12.0, 0, 0), (0, 6.0, 0), 6, 3, 1, 4.0)         # back wall (normal +z)
    mb.add_grid((-6.0, 0.0, 6.0), (0, 0, -12.0), (0, 6.0, 0), 6, 3, 1, 4.0)         # left wall (normal +x)
    pos, idx, nrm, tan, uv, mat = mb.arrays()
    a = np.stack([rng.uniform(-5, 5, num_slivers), rng.uniform(0.05, 4.0, num_slivers), rng.uniform(-5, 5, num_slivers)], -1)
    axis = rng.randint(0, 3, num_slivers)
    long_edge = np.zeros((num_slivers, 3)); long_edge[np.arange(num_slivers), axis] = rng.uniform(1.0, 6.0, num_slivers) * rng.choice([-1.0, 1.0], num_slivers)
    long_edge += rng.standard_normal((num_slivers, 3)) * (10.0 ** rng.uniform(-6, -3, num_slivers))[:, None]
    short_edge = rng.standard_normal((num_slivers, 3)); short_edge /= np.linalg.norm(short_edge, axis=1, keepdims=True)
    short_edge *= (10.0 ** rng.uniform(-5, -2, num_slivers))[:, None]
    p = np.stack([a, a + long_edge, a + short_edge], 1).reshape(-1, 3)
    n = np.cross(long_edge, short_edge); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    t = long_edge / np.linalg.norm(long_edge, axis=1, keepdims=True)
    base = len(pos)
    pos = np.concatenate([pos, p.astype(np.float32)]); nrm = np.concatenate([nrm, np.repeat(n, 3, 0).astype(np.float32)]); tan = np.concatenate([tan, np.repeat(t, 3, 0).astype(np.float32)])
    uv = np.concatenate([uv, np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (num_slivers, 1))])
    idx = np.concatenate([idx, (base + np.arange(3 * num_slivers, dtype=np.uint32)).reshape(-1, 3)]); mat = np.concatenate([mat, np.full(num_slivers, 2, dtype=np.uint32)])
    scene = ra.Scene()
    mats = [scene.add_material("diffuse", c) for c in ((0.7, 0.7, 0.7), (0.6, 0.5, 0.4), (0.8, 0.3, 0.2))]
    scene.add_mesh(pos, idx.astype(np.uint32), nrm, tan, uv, mat.astype(np.uint32), mats)
    scene.add_background_light((0.3, 0.4, 0.5))
    scene.add_directional_light((8.0, 7.5, 7.0), np.float32(0.0), ra.transform_from_euler((0.0, 0.0, 0.0), orientation))
    scene.build()
    return scene, ra.Camera((4.5, 2.5, 4.5), (15.0, 225.0, 0.0), aspect, 70.0)


@pytest.mark.parametrize("orientation", [(90.0, 0.0, 0.0), (0.0, 0.0, 0.0), (80.0, 0.0, 0.0)])
def test_axis_parallel_rays_among_sliver_triangles(built, walk, orientation):
    """Round-5 advisor: the counters-off binary walk skips boxes clearly off the fixed coordinate of an axis-parallel ray (boxNearDegenerateAxes, a 2^-7 margin),
    relying on two facts -- the builder's boxes bound their triangles' vertices in float, and Moeller-Trumbore accepts only points of the triangle.  An
    ill-conditioned sliver is where that could break.  6000 of them (extents down to 1e-5, long edges bent 1e-6 off an axis) under a delta sun straight down / along
    z / in a coordinate plane: the library's default walk (prune active) and the reference's counting walk (untouched) both give the oracle's image and ray
    counters -- so the pruned walk and the unpruned one agree on exactly the rays the argument is about."""
    w, h = 128, 80
    scene, camera = _sliver_mesh_under_a_delta_sun(w / h, orientation)
    out = run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=5)
    assert_identical(*out)
    assert out[2]["numShadowRays"] > out[2]["numShadowRaysHit"] > 0 and out[2]["numMeshHits"] > 0


def test_axis_parallel_ray_over_an_analytic_box_keeps_the_reference_s_bogus_hit(built, walk):
    """Round 6, found by the soak (tools/oracle_fuzz.py seed 9606, case 765; tools/oracle_fuzz_replay.py reproduces it): a diffuse bounce off an axis-aligned wall
    whose cosine sample returns a direction with an EXACTLY zero y component travels horizontally a metre above a BoxShape -- and the reference reports a hit on the
    box's top face: BoxShape::Intersect evaluates 0 * inf for the dropped axis and _mm_min_ps / _mm_max_ps keep the NaN's partner (the reference's own renderer,
    oracle/_ref/ref_render, gives this pixel 0 like the oracle).  Rounds 5's prune of boxes "clearly off an axis-parallel ray's fixed coordinate" skipped the
    top-level box and let the ray fly on (6 more path segments, a lit pixel).  The prune is sound for triangles (Moeller-Trumbore accepts only points of the
    triangle) and is applied inside meshes only now.  Both walks against the oracle, bit for bit, on the frame of the soak's case."""
    w, h = 128, 72
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=8000)
    out = run_both(scene, camera, w, h, walk=walk, passes=1, seed=677459682, max_ray_depth=10, min_russian_roulette_depth=4, dimensions=64, use_blue_noise=False)
    assert_identical(*out)
    assert out[3][21, 113].max() == 0.0 and out[0][21, 113].max() == 0.0      # the pixel whose path ends on the box it never touched


def test_mesh_two_level_bvh_bit_exact(built, walk):
    """Triangle mesh instance + analytic instances: mesh BVH traversal, Moller-Trumbore, barycentric frames."""
    w, h = 160, 90
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=20000)
    out = run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=8)
    assert_identical(*out)
    assert (out[2]["numRayTriangleTests"] > 0) == (walk == "counting") and out[2]["numMeshHits"] > 0


def test_device_texture_decode_matches_the_reference_vectors(built):
    """Every record of tests/golden/texture_kat.bin (BitmapTexture::Evaluate / CheckerboardTexture::Evaluate outputs of the
    REFERENCE: all 23 texel formats x colour spaces x filters, wrap and texel-edge coordinates, noise and (nested) mix
    textures) evaluated ON THE DEVICE through
    rtgpu_evaluate_textures: bit-exact."""
    import kat_io
    k = kat_io.load_texture_kat()
    lib = ra.rtgpu_lib()
    ctx = C.c_void_p()
    assert lib.rtgpu_create(0, C.byref(ctx)) == 0
    d = ra.RtSceneDesc()
    d.abiVersion = lib.rtgpu_abi_version()
    d.numTextures = len(k["textures"])
    d.textures = k["textures"]
    d.texelData = k["texels"].ctypes.data
    d.texelBytes = k["texels"].size
    assert lib.rtgpu_upload_scene(ctx, C.byref(d)) == 0, lib.rtgpu_last_error()
    ev = k["evals"]
    idx = np.ascontiguousarray(ev["texture"]); uv = np.ascontiguousarray(ev["uv"]); out = np.zeros((len(ev), 4), dtype=np.float32)
    assert lib.rtgpu_evaluate_textures(ctx, C.c_uint32(len(ev)), idx.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p),
                                       out.ctypes.data_as(C.c_void_p)) == 0, lib.rtgpu_last_error()
    bad = np.any(out.view(np.uint32) != np.ascontiguousarray(ev["out"]).view(np.uint32), axis=1)
    assert not bad.any(), [(int(ev["texture"][i]), k["textures"][int(ev["texture"][i])].format, ev["uv"][i], out[i], ev["out"][i]) for i in np.nonzero(bad)[0][:5]]
    lib.rtgpu_destroy(ctx)


def test_textured_materials_normal_maps_and_environment_map(built, walk):
    """bitmap textures (7 texel formats, sRGB and linear, the three filters) on base colour /
    roughness / metalness / emission, bitmap and procedural normal maps, checkerboard / const textures and an HDR
    environment map on the background light -- bit-exact against the oracle, which is itself pinned to the
    reference's BitmapTexture / Bitmap / Material code by tests/golden/texture_kat.bin."""
    w, h = 160, 90
    scene, camera = scene_zoo.textured_scene(w / h)
    d = scene.desc.contents
    assert d.numTextures == 13 and d.texelBytes > 0   # 8 bitmaps (one BC1), checkerboard, const, noise, mix, environment map
    out = run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=6)
    assert_identical(*out)
    # the textures are really in play: the same scene without them renders differently
    plain, cam2 = scene_zoo.mesh_scene(w / h, triangles=6000)
    vp = ra.Viewport(w, h, seed=99, max_ray_depth=6)
    vp.set_renderer(plain)
    vp.render(cam2, 3)
    assert not np.array_equal(vp.sum_buffer(), out[0])


@pytest.mark.parametrize("simple_bitmaps", [True, False])
def test_textured_sponza_class_scene_bit_exact(built, walk, simple_bitmaps):
    """The shading kernels are specialised per scene class (rt_device_core.h: lean / textured): a mesh-only, diffuse-only scene WITH albedo and
    normal maps on every material and an HDR environment map on the background light (bench.py --workload sponza-textured in small) runs
    the "lean + simple bitmaps" variant (8-bit BGRA and half-float RGBA bitmaps only: their evaluation is inlined) or, with one more texture
    of another kind in the scene, the "lean + textures" variant; the Cornell box and the all-BSDF scenes above run "anything without
    textures", the mesh scenes "lean", the textured zoo scene "anything"."""
    w, h = 96, 54
    scene, camera = scenes.sponza_class(w / h, 6000, textured=True, extra_texture=not simple_bitmaps)
    assert scene.desc.contents.numTextures == (17 if simple_bitmaps else 18)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=5))


def test_simple_bitmap_formats_and_filters_bit_exact(built):
    """Scene class "lean + simple bitmaps": every format the inlined evaluation decodes (24-bit BGR as Demo/MeshLoader.cpp's .bmp files, BGRA8,
    RGBA8, half-float RGBA) under the three filters, sRGB and linear, as albedo / normal / environment maps."""
    w, h = 64, 48
    pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(3000, 11, refine=True)
    rng = np.random.RandomState(5)
    scene = ra.Scene()
    mats = [scene.add_material("diffuse", c) for _, c in scenes.SPONZA_MATERIALS]
    formats = [("B8G8R8_UNorm", 3, np.uint8), ("B8G8R8A8_UNorm", 4, np.uint8), ("R8G8B8A8_UNorm", 4, np.uint8), ("R16G16B16A16_Half", 4, np.float16)]
    filters = ["nearest", "bilinear", "smoothstep"]
    for i, m in enumerate(mats):
        name, channels, dtype = formats[i % len(formats)]
        size = (37 + 5 * i, 64 + 3 * i)   # odd sizes: the wrap of the secondary texel
        texels = (rng.uniform(0.05, 1.0, size=size + (channels,)) * (255 if dtype == np.uint8 else 1)).astype(dtype)
        scene.set_material_texture(m, "baseColor", scene.add_bitmap_texture(texels, name, linear_space=(i % 2 == 0), filter=filters[i % 3]))
        bump = (127.5 + 25.0 * rng.standard_normal(size + (4,))).clip(0, 255).astype(np.uint8)
        scene.set_material_texture(m, "normal", scene.add_bitmap_texture(bump, "R8G8B8A8_UNorm", filter=filters[(i + 1) % 3]), 0.8)
    env = scene.add_bitmap_texture(rng.uniform(0.2, 1.5, size=(32, 64, 4)).astype(np.float16), "R16G16B16A16_Half", filter="bilinear")
    scene.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
    scene.add_background_light((1.0, 1.5, 2.0), texture=env)
    scene.build()
    camera = ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), w / h, 65.0)
    assert_identical(*run_both(scene, camera, w, h, walk="default", passes=2, max_ray_depth=4))


def test_ingested_json_obj_scene_bit_exact(built, walk):
    """SURVEY 8(f) row 3 end to end: a JSON scene file (helpers::LoadScene) with an OBJ mesh + MTL materials + BMP texture
    (helpers::LoadMesh), textured materials, three light types and a depth-of-field camera, rendered on the GPU and
    by the oracle from the same flattened scene."""
    import os
    import kat_io
    w, h = 160, 100
    obj_dir = os.path.join(kat_io.GOLDEN, "obj")
    camera = ra.Camera()
    scene = ra.Scene().load_json(os.path.join(obj_dir, "scene.json"), data_path=obj_dir + "/", camera=camera)
    scene.build()
    camera.set_perspective(w / h, np.float32(48.0) / np.float32(180.0) * np.float32(3.14159265359))
    d = scene.desc.contents
    assert d.numMeshes == 1 and d.numTriangles > 200 and d.numTextures == 3 and d.numLights == 3   # checkerboard + the BMP twice (scene file, MTL)
    out = run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=5)
    assert_identical(*out)
    assert out[2]["numMeshHits"] > 0 and out[2]["numAnalyticHits"] > 0


def test_single_object_scene_bypasses_top_bvh(built, walk):
    """Sponza-class configuration: ONE object (Scene::Traverse bypasses the BVH, Scene.cpp:231-235), two global lights."""
    w, h = 128, 72
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=8000, with_analytic=False)
    assert scene.desc.contents.numObjects == 1
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=8))


def test_lean_and_generic_shade_variants_agree(built, monkeypatch):
    """Scenes with only meshes / diffuse materials / background + directional lights run a feature-specialised shade
    kernel; it must produce exactly what the generic kernel produces (RTGPU_NO_LEAN forces the generic one)."""
    w, h = 128, 72
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=8000, with_analytic=False)
    images = []
    for no_lean in (False, True):
        if no_lean:
            monkeypatch.setenv("RTGPU_NO_LEAN", "1")
        else:
            monkeypatch.delenv("RTGPU_NO_LEAN", raising=False)
        vp = ra.Viewport(w, h, seed=11, max_ray_depth=8)
        vp.set_renderer(scene)
        vp.render(camera, 3)
        images.append((vp.sum_buffer(), vp.counters()))
    assert np.array_equal(images[0][0].view(np.uint32), images[1][0].view(np.uint32))
    assert images[0][1] == images[1][1]


def test_pass_batching_is_invisible(built, monkeypatch):
    """Up to RTGPU_PASS_BATCH passes share one launch sequence; per-pixel accumulation stays in pass order, so any
    batch size gives the same bits (5 passes: full batches + a partial one)."""
    w, h = 96, 72
    scene, camera = scenes.cornell_box(w / h)
    results = []
    for batch in ("1", "2", "8"):
        monkeypatch.setenv("RTGPU_PASS_BATCH", batch)
        vp = ra.Viewport(w, h, seed=21, max_ray_depth=4)
        vp.set_renderer(scene)
        vp.render(camera, 5)
        results.append((vp.sum_buffer(secondary=True), vp.counters()))
    for r in results[1:]:
        assert np.array_equal(r[0][0].view(np.uint32), results[0][0][0].view(np.uint32))
        assert np.array_equal(r[0][1].view(np.uint32), results[0][0][1].view(np.uint32))
        assert r[1] == results[0][1]


def test_batch_lanes_are_invisible(built, monkeypatch):
    """Consecutive batches run on alternating streams with their own path arenas (rtgpu_set_concurrency); the film is
    still summed in pass order, so 1..6 lanes give the same bits.  7 passes at batch size 2 = 4 batches in flight."""
    w, h = 96, 72
    scene, camera = scenes.cornell_box(w / h)
    monkeypatch.setenv("RTGPU_PASS_BATCH", "2")
    results = []
    for lanes in (1, 2, 3, 6):
        vp = ra.Viewport(w, h, seed=33, max_ray_depth=4)
        vp.set_renderer(scene)
        assert ra.rtgpu_lib().rtgpu_set_concurrency(vp.device_context(), lanes) == 0
        vp.render(camera, 7)
        results.append((vp.sum_buffer(secondary=True), vp.counters()))
    for r in results[1:]:
        assert np.array_equal(r[0][0].view(np.uint32), results[0][0][0].view(np.uint32))
        assert np.array_equal(r[0][1].view(np.uint32), results[0][0][1].view(np.uint32))
        assert r[1] == results[0][1]
    assert ra.rtgpu_lib().rtgpu_set_concurrency(vp.device_context(), 7) == -1   # RTGPU_ERR_INVALID_ARGUMENT


def test_lane_memory_budget_is_invisible(built, monkeypatch):
    """RTGPU_LANE_BUDGET_MB (a co-tenant's knob) bounds the device memory a batch lane takes for its path-state arenas; the batch a
    lane holds shrinks to fit.  640x360 needs ~95 MB per pass and lane, so 128 MB means one pass per launch where the default streams 20: same bits, same counters."""
    w, h = 640, 360
    scene, camera = scenes.sponza_class(w / h, 20000)
    results = []
    for budget in (None, "128"):
        if budget:
            monkeypatch.setenv("RTGPU_LANE_BUDGET_MB", budget)
        vp = ra.Viewport(w, h, seed=91, max_ray_depth=5)
        vp.set_renderer(scene)
        vp.render(camera, 7)
        results.append((vp.sum_buffer(secondary=True), vp.counters()))
    assert np.array_equal(results[1][0][0].view(np.uint32), results[0][0][0].view(np.uint32))
    assert np.array_equal(results[1][0][1].view(np.uint32), results[0][0][1].view(np.uint32))
    assert results[1][1] == results[0][1]


def test_front_buffer_postprocess_matches_oracle(built):
    """Viewport::PostProcessTile on the device (rtgpu_postprocess) against the oracle's restatement (itself pinned to the
    reference's FastLog / FastExp / ToneMap / ToBGR by postprocess_kat.bin) on a rendered sum buffer: identical 8-bit
    pixels for all four tone mappers, with and without dithering (same per-pixel hash on both sides)."""
    w, h = 160, 120
    scene, camera = scenes.cornell_box(w / h)
    vp = ra.Viewport(w, h, seed=5, max_ray_depth=4)
    vp.set_renderer(scene)
    vp.render(camera, 6)
    sum_buffer = np.ascontiguousarray(vp.sum_buffer())
    o = oracle_lib.lib()
    for tonemapper in (0, 1, 2, 3):
        for dithering, exposure, contrast, saturation in ((0.0, 0.0, 0.8, 0.98), (0.005, 1.3, 1.1, 0.5)):
            got = vp.front_buffer(exposure=exposure, contrast=contrast, saturation=saturation, dithering=dithering, tonemapper=tonemapper,
                                  color_filter=(1.0, 0.9, 0.8, 1.0), dither_seed=17)
            p = ra.RtPostprocessParams()
            p.colorFilter[0], p.colorFilter[1], p.colorFilter[2], p.colorFilter[3] = 1.0, 0.9, 0.8, 1.0
            p.exposure, p.contrast, p.saturation, p.ditheringStrength, p.bloomFactor = exposure, contrast, saturation, dithering, 0.0
            p.tonemapper, p.numPasses, p.ditherSeed = tonemapper, 6, 17
            ref = np.zeros((h, w), dtype=np.uint32)
            o.rto_postprocess(sum_buffer.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.byref(p), ref.ctypes.data_as(C.c_void_p))
            assert np.array_equal(got, ref), (tonemapper, dithering, int((got != ref).sum()))
            assert got.max() > 0x202020 and len(np.unique(got)) > 100
    p.bloomFactor = 0.5
    out = np.zeros((h, w), dtype=np.uint32)
    assert ra.rtgpu_lib().rtgpu_postprocess(vp.device_context(), C.byref(p), out.ctypes.data_as(C.c_void_p)) == -6   # RTGPU_ERR_UNSUPPORTED


def test_many_lights_all_strategy_and_degenerate_sizes(built, walk):
    """Edge cases: 16 lights under LightSamplingStrategy::All (requests past the eighth use the per-lane queue append, black
    lights produce no shadow ray), a one-triangle mesh without normals / tangents / uvs, a 1 x 1 viewport, a viewport that is
    not a multiple of the 64-pixel tiles or the 8 x 8 blocks, maxRayDepth = 0 (direct light only) and Russian roulette
    from the first bounce."""
    for (w, h, depth, rr) in ((67, 41, 5, 0), (1, 1, 3, 1), (96, 64, 0, 1)):
        scene, camera = scene_zoo.many_lights_scene(w / h)
        assert scene.desc.contents.numLights == 16
        out = run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=depth, min_russian_roulette_depth=rr, dimensions=256, light_sampling_all=True)
        assert_identical(*out)
        assert out[2]["numShadowRays"] > 0


def test_sixty_four_lights_at_full_hd_under_the_all_strategy(built):
    """LightSamplingStrategy::All makes path slots fat (two records per light): 64 lights at 1920x1080 would be ~109 GB per batch lane at the
    streaming batch size.  The lane budget knows the light count (bytesPerSlot / maxBatchFor): the passes run in smaller batches instead
    of failing to allocate, and 1/48 of the frame's tiles equal the oracle bit for bit."""
    w, h = 1920, 1080
    scene, camera = scene_zoo.many_lights_scene(w / h, num_point_lights=62)
    assert scene.desc.contents.numLights == 64
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=8, max_ray_depth=3, light_sampling_all=True, dimensions=512)
    vp.set_renderer(scene)
    ref = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    shard = (5, 48)
    for _ in range(3):
        p = vp.next_pass_params(camera)
        vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, None, cnt, shard=shard, threads=32)
    img = vp.sum_buffer()
    c = vp.counters()
    assert np.isfinite(img).all() and c["numShadowRays"] > 10 * c["numPrimaryRays"]
    owned = ref.any(axis=2)
    assert owned.sum() > 20000
    assert np.array_equal(img[owned].view(np.uint32), ref[owned].view(np.uint32))


def test_reset_restarts_the_accumulation(built):
    """rtgpu_reset (Viewport::Reset) zeroes sums and counters; the passes rendered afterwards with the same constants give
    the bits a fresh viewport gives."""
    w, h = 96, 72
    scene, camera = scenes.cornell_box(w / h)
    vp = ra.Viewport(w, h, seed=3, max_ray_depth=4)
    vp.set_renderer(scene)
    params = [vp.next_pass_params(camera) for _ in range(3)]
    for p in params:
        vp.render_pass_with(p)
    first = vp.sum_buffer().copy(); c_first = vp.counters()
    ra.host_lib().rth_viewport_reset(vp._h)
    assert not vp.sum_buffer().any() and vp.counters()["numRays"] == 0
    for p in params:
        vp.render_pass_with(p)
    assert np.array_equal(vp.sum_buffer().view(np.uint32), first.view(np.uint32)) and vp.counters() == c_first


def _oracle_block_error(sum_buffer, secondary, passes, block):
    o = oracle_lib.lib()
    o.rto_block_error.restype = C.c_float
    h, w = sum_buffer.shape[:2]
    return o.rto_block_error(sum_buffer.ctypes.data_as(C.c_void_p), secondary.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.c_uint32(passes),
                             C.c_uint32(block[0]), C.c_uint32(block[1]), C.c_uint32(block[2]), C.c_uint32(block[3]))


def test_block_errors_and_average_error_match_oracle(built):
    """Viewport::ComputeBlockError on the device (row-wise then block-wise sums in the reference's order) against the oracle's
    restatement on the same sum buffers: bit-identical floats for arbitrary blocks and for the whole image (averageError)."""
    w, h = 200, 136
    scene, camera = scenes.cornell_box(w / h)
    vp = ra.Viewport(w, h, seed=7, max_ray_depth=4)
    vp.set_renderer(scene)
    vp.render(camera, 6)
    s, s2 = vp.sum_buffer(secondary=True)
    s, s2 = np.ascontiguousarray(s), np.ascontiguousarray(s2)
    blocks = [(0, w, 0, h), (0, 64, 0, 64), (64, 200, 10, 11), (199, 200, 135, 136), (3, 130, 50, 136)]
    arr = (ra.RtBlock * len(blocks))(*[ra.RtBlock(*b) for b in blocks])
    out = (C.c_float * len(blocks))()
    assert ra.rtgpu_lib().rtgpu_compute_block_errors(vp.device_context(), C.c_uint32(6), C.c_uint32(len(blocks)), arr, out) == 0
    for b, got in zip(blocks, out):
        ref = _oracle_block_error(s, s2, 6, b)
        assert np.float32(got).view(np.uint32) == np.float32(ref).view(np.uint32), (b, got, ref)
    assert vp.progress()["averageError"] == out[0] > 0.0
    bad = (ra.RtBlock * 1)(ra.RtBlock(0, w + 1, 0, h))
    assert ra.rtgpu_lib().rtgpu_compute_block_errors(vp.device_context(), C.c_uint32(6), C.c_uint32(1), bad, out) == -1


def test_adaptive_rendering_follows_the_reference_block_logic(built):
    """RenderingParams::adaptiveSettings through the mirror's Viewport: after the initial passes, every second pass the blocks
    are re-evaluated (converged ones dropped, half-converged ones split) and only active blocks keep receiving samples.  The
    image must equal what the oracle produces when it is fed the same pass constants and the same per-pass pixel masks, and the
    block list must equal a Python replay of UpdateBlocksList driven by oracle block errors."""
    w, h = 128, 96
    scene, camera = scenes.cornell_box(w / h)
    desc = scene.desc
    bn = ra.load_blue_noise(); desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=14, max_ray_depth=3)      # (a seed whose 12 passes show both a split and a drop)
    vp.set_renderer(scene)
    settings = dict(num_initial_passes=4, min_block_size=8, max_block_size=64, subdivision_treshold=0.35, convergence_treshold=0.12)
    vp.set_adaptive(True, **settings)
    blocks = [(x, min(w, x + 64), y, min(h, y + 64)) for y in range(0, h, 64) for x in range(0, w, 64)]
    assert vp.progress()["blocks"] == blocks
    ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32); cnt = np.zeros(16, dtype=np.uint64)
    saw_split = saw_drop = False
    for n in range(1, 13):
        p = vp.next_pass_params(camera)
        mask = np.zeros((h, w), dtype=bool)
        for (x0, x1, y0, y1) in blocks:
            mask[y0:y1, x0:x1] = True
        # oracle pass into scratch buffers, then keep only the active pixels
        t = np.zeros_like(ref); t2 = np.zeros_like(ref)
        oracle_lib.render_pass(desc, p, w, h, t, t2, None, threads=8)
        ref[mask] += t[mask]; ref2[mask] += t2[mask]
        if blocks:
            vp.render_pass_with(p)      # = RenderPass + FinishPass (which runs UpdateBlocksList after every second pass)
        else:
            ra.host_lib().rth_viewport_finish_pass(vp._h)   # the mirror's Render() submits nothing once every block has converged
        if n % 2 == 0 and n >= settings["num_initial_passes"]:
            new_blocks, i, cur = [], 0, list(blocks)
            while i < len(cur):
                b = cur[i]
                err = _oracle_block_error(ref, ref2, n, b)
                bw, bh = b[1] - b[0], b[3] - b[2]
                if err < settings["convergence_treshold"]:
                    cur[i] = cur[-1]; cur.pop(); saw_drop = True; i += 1; continue
                if err < settings["subdivision_treshold"] and (bw > settings["min_block_size"] or bh > settings["min_block_size"]):
                    cur[i] = cur[-1]; cur.pop(); saw_split = True
                    if bw > bh:
                        half = (b[0] + b[1]) // 2; new_blocks += [(b[0], half, b[2], b[3]), (half, b[1], b[2], b[3])]
                    else:
                        half = (b[2] + b[3]) // 2; new_blocks += [(b[0], b[1], b[2], half), (b[0], b[1], half, b[3])]
                i += 1
            blocks = cur + new_blocks
        assert vp.progress()["blocks"] == blocks, n
    img, img2 = vp.sum_buffer(secondary=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(img2.view(np.uint32), ref2.view(np.uint32))
    assert saw_split and saw_drop


def test_depth_of_field(built, walk):
    w, h = 96, 72
    scene, camera = scenes.cornell_box(w / h)
    camera.set_dof(True, 11.0, 0.3)
    assert_identical(*run_both(scene, camera, w, h, walk=walk, passes=3, max_ray_depth=3))
    # the other bokeh shapes of Camera::GenerateBokeh (hexagon: the reference only ever samples its first rhombus; square) and barrel
    # distortion (its random factor comes from the per-pixel generator); camera_ray.kat pins all of them to the reference
    for shape in (1, 2):
        camera.set_lens(bokeh_shape=shape, barrel_const=0.02, barrel_variable=0.03)
        out = run_both(scene, camera, w, h, walk=walk, passes=2, max_ray_depth=3)
        assert_identical(*out)


def test_empty_scene_and_background_only(built, walk):
    """RenderingTest.EmptyScene / BackgroundLightOnly (Tests/RaytracingTests.cpp:263-315) with their tolerances."""
    w = h = 32
    scene = ra.Scene().build()
    cam = ra.Camera((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.0, 90.0)
    img, _, counters, ref, _, _, _ = run_both(scene, cam, w, h, walk=walk, passes=1)
    assert np.all(img == 0.0) and np.all(ref == 0.0)
    scene = ra.Scene()
    scene.add_background_light((1.0, 2.0, 3.0))
    scene.build()
    out = run_both(scene, cam, w, h, walk=walk, passes=1)
    assert_identical(*out)
    assert np.all(np.abs(out[0] - np.array([1.0, 2.0, 3.0], dtype=np.float32)) <= 0.01)


@pytest.mark.parametrize("bsdf,passes,expected,tol,kwargs", [
    ("diffuse", 100, (0.4, 1.2, 2.4), 0.05, {}),
    ("null", 1, (3.0, 2.0, 1.0), 0.0, {"base_color": (0.0, 0.0, 0.0), "emission": (3.0, 2.0, 1.0)}),
    ("metal", 20, (0.4, 1.2, 2.4), 0.05, {"ior": 0.0, "k": 100.0}),
    ("dielectric", 1000, (1.0, 2.0, 3.0), 0.075, {"base_color": (1.0, 1.0, 1.0)}),
])
def test_reference_furnace_tests_on_gpu(built, bsdf, passes, expected, tol, kwargs):
    """The reference's own RenderingTest.FurnaceTest_* (Tests/RaytracingTests.cpp:317-523): every pixel of the
    sum buffer / numPasses within the reference's tolerance, 32x32 viewport, default RenderingParams."""
    w = h = 32
    scene, camera = scenes.furnace(bsdf, **kwargs)
    vp = ra.Viewport(w, h, seed=2024)
    vp.set_renderer(scene)
    vp.render(camera, passes)
    img = vp.sum_buffer() / np.float32(passes)
    assert np.all(np.abs(img - np.array(expected, dtype=np.float32)) <= tol + 1e-6), float(np.abs(img - np.array(expected)).max())


def test_tile_sharding_reproduces_single_gpu_image(built):
    """64x64-tile interleaved ownership: the sum of the shard images equals the unsharded image bit for bit."""
    w, h = 200, 136
    scene, camera = scenes.cornell_box(w / h)
    full = ra.Viewport(w, h, seed=5, max_ray_depth=4)
    full.set_renderer(scene)
    params = [full.next_pass_params(camera) for _ in range(2)]
    for p in params:
        full.render_pass_with(p)
    whole = full.sum_buffer()
    total = np.zeros_like(whole)
    rays = 0
    for rank in range(3):
        vp = ra.Viewport(w, h, seed=5, max_ray_depth=4)
        vp.set_renderer(scene)
        vp.set_shard(rank, 3)
        for p in params:
            vp.render_pass_with(p)
        part = vp.sum_buffer()
        assert np.all((part == 0) | (total == 0))   # disjoint support
        total += part
        rays += vp.counters()["numRays"]
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))
    assert rays == full.counters()["numRays"]


def test_c_abi_error_paths(built):
    lib = ra.rtgpu_lib()
    ctx = C.c_void_p()
    assert lib.rtgpu_create(0, C.byref(ctx)) == 0
    p = ra.RtPassParams()
    assert lib.rtgpu_render_pass(ctx, C.byref(p)) == -5          # RTGPU_ERR_NOT_READY: no scene yet
    assert b"upload_scene" in lib.rtgpu_last_error()
    assert lib.rtgpu_resize(ctx, 0, 10) == -1                     # invalid size
    d = ra.RtSceneDesc()
    d.abiVersion = 999
    assert lib.rtgpu_upload_scene(ctx, C.byref(d)) == -1
    lib.rtgpu_destroy(ctx)
    assert lib.rtgpu_create(99, C.byref(ctx)) == -1               # device index out of range


def test_walk_info_names_the_kernel_that_serves_the_scene(built):
    """rtgpu_get_walk_info: single-mesh scenes walk the 4-wide collapse, mesh + analytic scenes the two-level one, either falls back to the
    reference's binary tree with the intersection counters on; the byte counts are those of the uploaded arrays."""
    class WalkInfo(C.Structure):
        _fields_ = [("kernel", C.c_uint32), ("reserved", C.c_uint32), ("nodeBytes", C.c_uint64), ("leafBoxBytes", C.c_uint64), ("triangleBytes", C.c_uint64)]
    lib = ra.rtgpu_lib()
    for make, expected in ((lambda a: scenes.sponza_class(a, 6000), 1), (lambda a: scene_zoo.mesh_scene(a, triangles=2000), 2), (scenes.sphere_area_light, 0)):
        scene, camera = make(1.0)
        vp = ra.Viewport(32, 32, seed=1)
        vp.set_renderer(scene)
        vp.render(camera, 1)   # (the mirror uploads the scene with the first pass)
        wi = WalkInfo()
        assert lib.rtgpu_get_walk_info(vp.device_context(), C.byref(wi)) == 0
        assert wi.kernel == expected, (wi.kernel, expected)
        d = scene.desc.contents
        assert wi.triangleBytes == d.numTriangles * 36
        if expected == 0:
            assert wi.nodeBytes == (d.numTopNodes + d.numMeshNodes) * 32
        else:
            assert wi.nodeBytes % 64 == 0 and wi.nodeBytes > 0 and wi.leafBoxBytes > 0
            assert lib.rtgpu_set_intersection_counters(vp.device_context(), 1) == 0
            assert lib.rtgpu_get_walk_info(vp.device_context(), C.byref(wi)) == 0 and wi.kernel == 0
    assert lib.rtgpu_get_walk_info(None, None) == -1


def test_full_size_sponza_class_properties(built, walk):
    """BASELINE config 3 at its full size (1920x1080, Sponza-class mesh of ~262 k triangles, depth 8), through
    size-independent properties and an oracle-checked sample -- with walk = "default" this is exactly the pipeline bench.py times
    (k_trace_wide + its re-trace hand-over, dense path state, four batch lanes, intersection counters off):
      * the tiles one shard owns (1/32 of the frame, spread over the whole image) equal the CPU oracle bit for bit,
        with identical counters -- the GPU renders exactly those tiles via rtgpu_set_shard;
      * the 2-shard images are disjoint and add up to the unsharded image bit for bit;
      * counters: numPrimaryRays = pixels x passes, numRays >= numPrimaryRays, numShadowRaysHit <= numShadowRays,
        and the traced-ray tally is the same with one lane and with three (batches overlap)."""
    w, h, depth, passes = 1920, 1080, 8, 3
    scene, camera = scenes.sponza_class(w / h)
    assert abs(scene.desc.contents.numTriangles - 262144) <= 2621
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data

    full = ra.Viewport(w, h, seed=4242, max_ray_depth=depth)
    full.set_renderer(scene, intersection_counters=(walk == "counting"))
    params = [full.next_pass_params(camera) for _ in range(passes)]
    for p in params:
        full.render_pass_with(p)
    whole = full.sum_buffer()
    cw = full.counters()
    assert np.isfinite(whole).all() and float(whole.max()) > 0.0
    assert cw["numPrimaryRays"] == w * h * passes
    assert cw["numRays"] >= cw["numPrimaryRays"] and cw["numShadowRaysHit"] <= cw["numShadowRays"]
    assert cw["numMeshHits"] > 0 and cw["numAnalyticHits"] == 0

    # oracle-checked sample: shard 0 of 32
    part_vp = ra.Viewport(w, h, seed=4242, max_ray_depth=depth)
    part_vp.set_renderer(scene, intersection_counters=(walk == "counting"))
    part_vp.set_shard(0, 32)
    ref = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    for p in params:
        part_vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, None, cnt, shard=(0, 32), threads=16)
    part = part_vp.sum_buffer()
    assert np.array_equal(part.view(np.uint32), ref.view(np.uint32))
    pc = part_vp.counters()
    for i, n in enumerate(ra.COUNTER_NAMES):
        if n in (COMPARED if walk == "counting" else NOT_INTERSECTION):
            assert pc[n] == int(cnt[i]), (n, pc[n], int(cnt[i]))
    if walk == "default":
        assert cw["numRetracedRays"] > 0 and cw["numRayBoxTests"] == 0   # the 4-wide walk ran and handed its undecided rays over
    owned = part != 0
    assert np.array_equal(part[owned].view(np.uint32), whole[owned].view(np.uint32))   # the same pixels in the full frame

    # two shards: disjoint, exact sum; one lane
    total = np.zeros_like(whole)
    rays = 0
    for rank in range(2):
        vp = ra.Viewport(w, h, seed=4242, max_ray_depth=depth)
        vp.set_renderer(scene, intersection_counters=(walk == "counting"))
        vp.set_shard(rank, 2)
        assert ra.rtgpu_lib().rtgpu_set_concurrency(vp.device_context(), 1) == 0
        for p in params:
            vp.render_pass_with(p)
        img = vp.sum_buffer()
        assert np.all((img == 0) | (total == 0))
        total += img
        rays += vp.counters()["numRays"]
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))
    assert rays == cw["numRays"]


def test_full_size_sponza_class_under_the_all_strategy(built):
    """SURVEY 8(d)'s second C3 run -- LightSamplingStrategy::All with dimensions = 128, the configuration whose sample stream is path-exact
    (PathTracerMIS.cpp:141-147) -- at BASELINE's full size, with the library's defaults (4-wide walk, dense path state with one request per
    light and vertex): the tiles shard 0 of 32 owns equal the CPU oracle bit for bit, with identical ray counters, and they are the same pixels
    of the unsharded frame (bench.py --workload sponza-all times this pipeline)."""
    w, h, depth, passes = 1920, 1080, 8, 2
    scene, camera = scenes.sponza_class(w / h)
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    args = dict(seed=777, max_ray_depth=depth, dimensions=128, light_sampling_all=True)
    full = ra.Viewport(w, h, **args)
    full.set_renderer(scene)
    params = [full.next_pass_params(camera) for _ in range(passes)]
    for p in params:
        full.render_pass_with(p)
    whole = full.sum_buffer()
    cw = full.counters()
    assert cw["numPrimaryRays"] == w * h * passes and cw["numRetracedRays"] > 0 and cw["numShadowRays"] > cw["numRays"]   # two lights: more shadow rays than path segments
    part_vp = ra.Viewport(w, h, **args)
    part_vp.set_renderer(scene)
    part_vp.set_shard(0, 32)
    ref = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    for p in params:
        part_vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, None, cnt, shard=(0, 32), threads=16)
    part = part_vp.sum_buffer()
    assert np.array_equal(part.view(np.uint32), ref.view(np.uint32))
    pc = part_vp.counters()
    for i, n in enumerate(ra.COUNTER_NAMES):
        if n in NOT_INTERSECTION:
            assert pc[n] == int(cnt[i]), (n, pc[n], int(cnt[i]))
    owned = part != 0
    assert np.array_equal(part[owned].view(np.uint32), whole[owned].view(np.uint32))


def test_plain_path_tracer_bit_exact(built, walk):
    """Renderer "Path Tracer" (PathTracer.cpp: BSDF sampling only): all lights x all BSDFs, the mesh scene and the Cornell box
    against the oracle's restatement -- images and counters identical; no shadow rays are cast."""
    for scene, camera, w, h in ((lambda a: scene_zoo.all_lights_scene(a), None, 96, 72), (lambda a: scene_zoo.mesh_scene(a, triangles=20000), None, 96, 54),
                                (lambda a: scenes.cornell_box(a), None, 80, 60)):
        sc, cam = scene(w / h)
        desc = sc.desc
        bn = ra.load_blue_noise()
        desc.contents.blueNoise = bn.ctypes.data
        vp = ra.Viewport(w, h, seed=7, max_ray_depth=6)
        vp.set_renderer(sc, name="Path Tracer", intersection_counters=(walk == "counting"))
        ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32)
        cnt = np.zeros(16, dtype=np.uint64)
        for _ in range(3):
            p = vp.next_pass_params(cam)
            vp.render_pass_with(p)
            oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=8, plain=True)
        img, img2 = vp.sum_buffer(secondary=True)
        out = (img, img2, vp.counters(), ref, ref2, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES)}, walk)
        assert_identical(*out)
        assert out[2]["numShadowRays"] == 0 and out[2]["numRays"] > 3 * w * h


def test_debug_renderer_all_modes_bit_exact(built):
    """Renderer "Debug" (DebugRenderer.cpp): the 13 DebugRenderingMode values on the textured mesh scene (normal maps, texture
    coordinates, textured material parameters) and the all-lights scene (light hits are yellow, misses black) against the oracle's
    restatement; the TriangleID colouring (Hash -> HSVtoRGB) is pinned to the reference by debug_triangle_id.kat."""
    for make, w, h in ((scene_zoo.textured_scene, 96, 54), (scene_zoo.all_lights_scene, 80, 60)):
        scene, camera = make(w / h)
        desc = scene.desc
        bn = ra.load_blue_noise()
        desc.contents.blueNoise = bn.ctypes.data
        vp = ra.Viewport(w, h, seed=11)
        vp.set_renderer(scene, name="Debug")
        images = []
        for mode in range(13):
            vp.set_debug_mode(mode)
            vp.reset()
            ref = np.zeros((h, w, 3), dtype=np.float32)
            for _ in range(2):
                p = vp.next_pass_params(camera)
                vp.render_pass_with(p)
                oracle_lib.render_pass_debug(desc, p, w, h, mode, ref, threads=8)
            img = vp.sum_buffer()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), mode
            images.append(img.copy())
        assert len({im.tobytes() for im in images}) >= 11   # the modes really differ (two material parameters may coincide)


def test_frames_beyond_full_hd_stream_in_smaller_batches(built):
    """3840x2160 (8.3 M pixels): streaming grows the pass batch only to 16 (an arena stays below ~24 GB), and the shard-sampled
    pixels of the full frame equal the oracle bit for bit after 20 streamed passes' worth of batching logic (8 + 12: two launch
    sequences, the second one larger).  Cornell box, depth 4."""
    w, h, depth, passes = 3840, 2160, 4, 20
    scene, camera = scenes.cornell_box(w / h)
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    full = ra.Viewport(w, h, seed=77, max_ray_depth=depth)
    full.set_renderer(scene)
    params = [full.next_pass_params(camera) for _ in range(passes)]
    for p in params:
        full.render_pass_with(p)          # no synchronising call in between
    whole = full.sum_buffer()
    cw = full.counters()
    assert cw["numPrimaryRays"] == w * h * passes and np.isfinite(whole).all()
    ref = np.zeros((h, w, 3), dtype=np.float32)
    for p in params:
        oracle_lib.render_pass(desc, p, w, h, ref, None, None, shard=(5, 256), threads=16)
    owned = ref != 0
    assert owned.any()
    assert np.array_equal(whole[owned].view(np.uint32), ref[owned].view(np.uint32))


# ---- the default traversal of single-mesh scenes: the reference's tree re-encoded in 32-byte child pairs (rt_wide_grid.inl) -------
def run_quant(scene, camera, w, h, passes, seed=99, threads=8, shard=None, schedule=None, **vp_args):
    """Like run_both, with the intersection counters OFF (the reference's default): single-mesh scenes then run the 4-wide walk over the re-encoded
    ("quantised") tree, and what it does not trust is traced again by the binary-tree kernel."""
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=seed, **vp_args)
    vp.set_renderer(scene)
    if shard:
        vp.set_shard(*shard)
    assert ra.rtgpu_lib().rtgpu_set_intersection_counters(vp.device_context(), 0) == 0
    if schedule:   # (tail bounce, block-local re-trace): rtgpu_set_schedule
        for what, value in enumerate(schedule):
            assert ra.rtgpu_lib().rtgpu_set_schedule(vp.device_context(), C.c_uint32(what), C.c_int32(value)) == 0
    ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    for _ in range(passes):
        p = vp.next_pass_params(camera)
        vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, shard=shard or (0, 1), threads=threads)
    img, img2 = vp.sum_buffer(secondary=True)
    return img, img2, vp.counters(), ref, ref2, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES)}


def assert_quant_identical(img, img2, counters, ref, ref2, ref_counters):
    assert np.isfinite(ref).all()
    nbad = int(np.count_nonzero(img.view(np.uint32) != ref.view(np.uint32)))
    assert nbad == 0, "%d of %d sum-buffer values differ (max abs %.3e)" % (nbad, ref.size, float(np.abs(img - ref).max()))
    assert np.array_equal(img2.view(np.uint32), ref2.view(np.uint32))
    for n in NOT_INTERSECTION:
        assert counters[n] == ref_counters[n], (n, counters[n], ref_counters[n])
    assert counters["numRayBoxTests"] == 0 and counters["numRayTriangleTests"] == 0   # the reference's counters belong to its own walk


@pytest.mark.parametrize("dense", ["0", "1", "back-to-front", "any-hit-nearest-first"])
def test_wide_traversal_bit_exact_on_single_mesh_scenes(built, monkeypatch, dense):
    """k_trace_wide (4-wide collapse of the same tree, conservative 16-bit boxes, exact leaf gate, runner-up tracking, exact re-trace,
    stack-overflow hand-over) gives the reference's hits: images and ray / shadow-ray / hit counters identical to the oracle's binary-tree
    walk, with the dense and with the slot-per-pixel path state, and only a small fraction of the rays needs the exact re-trace.
    Round 6: any-hit rays walk the FARTHEST child they enter first by default (occlusion is an OR over the candidates: the order cannot
    change a result); the last case runs them nearest first like closest-hit rays (RTGPU_ANYHIT_FAR_FIRST=0) -- same bits either way."""
    monkeypatch.setenv("RTGPU_WIDE", "1")
    monkeypatch.setenv("RTGPU_NO_DENSE", "1" if dense == "0" else "0")
    if dense == "back-to-front":
        monkeypatch.setenv("RTGPU_WIDE_REVERSE", "1")   # (round 5's default: a launch's queue taken from its end; round 6 takes it front to back)
    if dense == "any-hit-nearest-first":
        monkeypatch.setenv("RTGPU_ANYHIT_FAR_FIRST", "0")
    w, h = 128, 72
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=8000, with_analytic=False)
    out = run_quant(scene, camera, w, h, passes=3, max_ray_depth=8)
    assert_quant_identical(*out)
    traced = out[2]["numRays"] + out[2]["numShadowRays"]
    assert 0 < out[2]["numRetracedRays"] < 0.02 * traced, (out[2]["numRetracedRays"], traced)
    scene, camera = scenes.sponza_class(w / h, 60000)
    out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=8, light_sampling_all=True, dimensions=128)
    assert_quant_identical(*out)
    out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=3, min_russian_roulette_depth=8)
    assert_quant_identical(*out)
    # a tree of a handful of nodes (a 12-triangle box, seen from inside and lit by the background through nothing): wide nodes with empty slots
    pos, idx, nrm, tan, uv, mat = scenes.box_mesh(1.0)
    box = ra.Scene()
    box.add_mesh(pos, idx, nrm, tan, uv, mat, [box.add_material("diffuse", (0.8, 0.3, 0.2)), box.add_material("diffuse", (0.3, 0.8, 0.2))])
    box.add_background_light((1.0, 1.5, 2.0))
    box.build()
    assert_quant_identical(*run_quant(box, ra.Camera((0.0, 2.5, 6.0), (20.0, 180.0, 0.0), w / h, 45.0), w, h, passes=2, max_ray_depth=4))


@pytest.mark.parametrize("after", ["1", "3", "12"])
def test_rays_handed_over_in_mid_walk_bit_exact(built, monkeypatch, after):
    """RTGPU_WIDE_DRAIN_ABORT=N: a wave of k_trace_wide whose work queue ran dry N loop iterations ago gives up the rays it still walks -- closest-hit rays
    whose best hit so far is already written through, any-hit rays half way -- and the re-trace launch (the reference's own walk) redoes them from the start:
    the stack-overflow hand-over taken by a large share of a small frame's rays instead of by none.  Images and counters stay the oracle's."""
    monkeypatch.setenv("RTGPU_WIDE", "1")
    monkeypatch.setenv("RTGPU_WIDE_DRAIN_ABORT", after)
    w, h = 128, 72
    scene, camera = scenes.sponza_class(w / h, 60000)
    for local_retrace in (0, 1):
        out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=8, schedule=(0, local_retrace))   # (no fused tail: every bounce through k_trace_wide)
        assert_quant_identical(*out)
        assert out[2]["numRetracedRays"] > 0
    scene, camera = scene_zoo.mesh_scene(w / h, triangles=8000, with_analytic=False)
    assert_quant_identical(*run_quant(scene, camera, w, h, passes=2, max_ray_depth=8, light_sampling_all=True, dimensions=128))


def test_fused_tail_and_block_local_retrace_are_invisible(built):
    """Round 4's launch-sequence variants (rtgpu_set_schedule): the fused tail kernel k_tail taking over at bounce 1, 2, 3, 5 or never, and the 4-wide
    walks tracing their undecided rays themselves or handing them to a launch of their own.  Every combination renders the oracle's image bit for
    bit with the oracle's ray counters -- BASELINE config 3's scene class (one mesh, background + directional light under `Single`: k_trace_wide,
    dense path state), several passes per batch, and the plain path tracer (no next event estimation) once."""
    w, h = 160, 90
    scene, camera = scenes.sponza_class(w / h, 20000)
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    lib = ra.rtgpu_lib()
    reference = None
    for tail_bounce, local_retrace in ((0, 0), (0, 1), (1, 1), (2, 0), (3, 1), (5, 0), (-1, -1)):
        vp = ra.Viewport(w, h, seed=4242, max_ray_depth=8)
        vp.set_renderer(scene)
        ctx = vp.device_context()
        assert lib.rtgpu_set_schedule(ctx, C.c_uint32(0), C.c_int32(tail_bounce)) == 0 and lib.rtgpu_set_schedule(ctx, C.c_uint32(1), C.c_int32(local_retrace)) == 0
        if reference is None:
            ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32)
            cnt = np.zeros(16, dtype=np.uint64)
            for _ in range(5):
                p = vp.next_pass_params(camera)
                vp.render_pass_with(p)
                oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=8)
            reference = (ref, ref2, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES)})
        else:
            vp.render(camera, 5)    # the same seed => the same per-pass constants
        img, img2 = vp.sum_buffer(secondary=True)
        counters = vp.counters()
        assert_quant_identical(img, img2, counters, *reference)
        assert counters["numRetracedRays"] > 0
    assert lib.rtgpu_set_schedule(ctx, C.c_uint32(2), C.c_int32(0)) == -1 and lib.rtgpu_set_schedule(ctx, C.c_uint32(1), C.c_int32(2)) == -1   # RTGPU_ERR_INVALID_ARGUMENT
    # the plain path tracer through the tail (k_tail<0, true>)
    plain = []
    for tail_bounce in (0, 2):
        vp = ra.Viewport(w, h, seed=4242, max_ray_depth=6)
        vp.set_renderer(scene, name="Path Tracer")
        assert lib.rtgpu_set_schedule(vp.device_context(), C.c_uint32(0), C.c_int32(tail_bounce)) == 0
        vp.render(camera, 3)
        plain.append((vp.sum_buffer(), vp.counters()))
    assert np.array_equal(plain[0][0].view(np.uint32), plain[1][0].view(np.uint32)) and plain[0][1] == plain[1][1]


@pytest.mark.parametrize("packets", ["0", "1"])
def test_packet_walk_of_the_camera_rays_gives_the_reference_hits(built, monkeypatch, packets):
    """rt_trace_packet.inl: the camera rays of a dense batch walk the 4-wide tree one 8 x 8 pixel block per wave (uniform node, shared stack, a child is
    entered if any lane's ray enters it).  Images and ray / shadow-ray / hit counters are the oracle's with it and without it (RTGPU_PACKET=0: k_trace_wide
    serves bounce 0 too): a frame whose last packet is ragged, a thin-lens camera (64 different origins per packet), a shard (packets of owned tiles only),
    and a batch of several passes."""
    monkeypatch.setenv("RTGPU_PACKET", packets)
    w, h = 150, 90     # 13 500 paths per pass: 210 full packets and one of 60
    scene, camera = scenes.sponza_class(w / h, 60000)
    out = run_quant(scene, camera, w, h, passes=3, max_ray_depth=6)
    assert_quant_identical(*out)
    camera.set_dof(True, 9.0, 0.25)
    out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=4)
    assert_quant_identical(*out)
    camera.set_dof(False)
    out = run_quant(scene, camera, 192, 128, passes=2, max_ray_depth=5, shard=(1, 3))
    assert_quant_identical(*out)


@pytest.mark.parametrize("abort_after", ["0", "3", None, "monsters"])
def test_retrace_launch_hands_long_closest_hit_rays_to_the_cooperative_walker(built, monkeypatch, abort_after):
    """Round 5: the re-trace launch behind k_trace_wide (the reference's own walk over the 0.1 % of the rays the 4-wide walk does not decide) CAN hand
    closest-hit rays that are still walking RT_ABORT_RETRACE_AFTER scheduling rounds after their wave's queue ran dry to k_trace_monster, which finds
    the same hit with a whole block (degenerate axis-parallel rays walk most of the tree: 1-1.6 ms alone in a wave).  The hand-over is OFF by default since
    the axis-parallel prune (rt_runtime.hip launchRetrace); setting RTGPU_ABORT_RETRACE_AFTER switches it on: 0 sends EVERY ray in flight at that moment down
    that path, 3 the slower ones.  The `None` case is the product default (no hand-over, no k_trace_monster launch) and RTGPU_RETRACE_MONSTERS=1 with the
    default threshold is covered by the fourth case.  Images and counters are the oracle's in all of them -- separate re-trace launch and block-local second
    walk (whose aborted rays go to the launch's exact queue, then re-trace launch, then monster), dense and slot-per-pixel path state."""
    if abort_after == "monsters":
        monkeypatch.setenv("RTGPU_RETRACE_MONSTERS", "1")
        abort_after = None
    if abort_after is not None:
        monkeypatch.setenv("RTGPU_ABORT_RETRACE_AFTER", abort_after)
    w, h = 160, 96
    scene, camera = scenes.sponza_class(w / h, 30000)
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    lib = ra.rtgpu_lib()
    reference = None
    for local_retrace, no_dense in ((0, False), (1, False), (0, True)):
        if no_dense:
            monkeypatch.setenv("RTGPU_NO_DENSE", "1")
        vp = ra.Viewport(w, h, seed=777, max_ray_depth=6)
        vp.set_renderer(scene)
        ctx = vp.device_context()
        assert lib.rtgpu_set_intersection_counters(ctx, 0) == 0
        assert lib.rtgpu_set_schedule(ctx, C.c_uint32(0), C.c_int32(0)) == 0 and lib.rtgpu_set_schedule(ctx, C.c_uint32(1), C.c_int32(local_retrace)) == 0
        if reference is None:
            ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32)
            cnt = np.zeros(16, dtype=np.uint64)
            for _ in range(3):
                p = vp.next_pass_params(camera)
                vp.render_pass_with(p)
                oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=8)
            reference = (ref, ref2, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES)})
        else:
            vp.render(camera, 3)    # the same seed => the same per-pass constants
        img, img2 = vp.sum_buffer(secondary=True)
        counters = vp.counters()
        assert_quant_identical(img, img2, counters, *reference)
        assert counters["numRetracedRays"] > 0


def test_wide_traversal_hands_over_rays_whose_stack_would_overflow(built, monkeypatch):
    """A pathological mesh -- 64 nested sheets around the camera axis, sizes and distances growing by 1.6 from one to the next: the SAH
    builder peels them off a few at a time (a tree 22 levels deep for 128 triangles), and a ray through the stack of sheets enters every
    child on its way down and defers more of them than a 24-entry stack holds (17 000 of 30 000 rays from the near side).  Such rays
    go to the binary-tree kernel (the uploaded depth picks its stack class); images and ray counters stay the oracle's, from both sides.
    The camera rays are the ones that overflow: they take k_trace_wide with RTGPU_PACKET=0 (the hand-over under test) and the packet walk
    of rt_trace_packet.inl by default (its shared stack holds them: no hand-over, the same image)."""
    n = 64
    pos, idx = [], []
    for k in range(n):
        r, z = 0.5 * 1.6 ** k, -2.0 - 1.6 ** k
        base = len(pos)
        pos += [(-r, -r, z), (r, -r, z), (r, r, z), (-r, r, z)]
        idx += [(base, base + 1, base + 2), (base, base + 2, base + 3)]
    pos = np.asarray(pos, dtype=np.float32); idx = np.asarray(idx, dtype=np.uint32)
    nrm = np.tile(np.asarray([[0.0, 0.0, 1.0]], dtype=np.float32), (len(pos), 1))
    tan = np.tile(np.asarray([[1.0, 0.0, 0.0]], dtype=np.float32), (len(pos), 1))
    uv = np.zeros((len(pos), 2), dtype=np.float32)
    scene = ra.Scene()
    m = scene.add_material("diffuse", (0.7, 0.6, 0.5))
    scene.add_mesh(pos, idx, nrm, tan, uv, np.zeros(len(idx), dtype=np.uint32), [m])
    scene.add_background_light((1.0, 1.5, 2.0))
    scene.build()
    w, h = 96, 64
    far = float(1.6 ** n)
    overflows = 0
    for packets, camera in [(p, c) for p in ("0", "1") for c in (ra.Camera((0.0, 0.0, 3.0), (0.0, 180.0, 0.0), w / h, 50.0), ra.Camera((0.0, 0.0, -3.0 * far), (0.0, 0.0, 0.0), w / h, 50.0))]:
        monkeypatch.setenv("RTGPU_PACKET", packets)
        out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=4)
        assert_quant_identical(*out)
        print("stack overflows: %d, untrusted: %d, re-traced: %d of %d rays" % (out[2]["numStackOverflowRays"], out[2]["numUntrustedRays"], out[2]["numRetracedRays"],
                                                                                 out[2]["numRays"] + out[2]["numShadowRays"]))
        overflows += out[2]["numStackOverflowRays"]
    assert overflows > 0


def test_page_locked_read_back_buffers(built):
    """rtgpu_host_register / rtgpu_host_unregister: a registered buffer receives the same frame as a pageable one; unregistering an unknown
    pointer is an error code, not a crash; NULL arguments are refused."""
    lib = ra.rtgpu_lib()
    w, h = 160, 96
    scene, camera = scenes.cornell_box(w / h)
    vp = ra.Viewport(w, h, seed=3, max_ray_depth=4)
    vp.set_renderer(scene)
    vp.render(camera, 3)
    ctx = vp.device_context()
    plain = np.zeros((h, w, 3), dtype=np.float32)
    pinned = np.zeros((h, w, 3), dtype=np.float32)
    assert lib.rtgpu_host_register(ctx, pinned.ctypes.data_as(C.c_void_p), C.c_size_t(pinned.nbytes)) == 0
    try:
        assert lib.rtgpu_read_sum(ctx, plain.ctypes.data_as(C.POINTER(C.c_float)), None) == 0
        assert lib.rtgpu_read_sum(ctx, pinned.ctypes.data_as(C.POINTER(C.c_float)), None) == 0
        assert plain.any() and np.array_equal(plain.view(np.uint32), pinned.view(np.uint32))
        assert np.array_equal(plain.view(np.uint32), vp.sum_buffer().view(np.uint32))        # the mirror's own (registered) bitmap
    finally:
        assert lib.rtgpu_host_unregister(ctx, pinned.ctypes.data_as(C.c_void_p)) == 0
    assert lib.rtgpu_host_unregister(ctx, plain.ctypes.data_as(C.c_void_p)) != 0              # never registered
    assert lib.rtgpu_host_register(ctx, None, C.c_size_t(16)) == -1 and lib.rtgpu_host_register(None, plain.ctypes.data_as(C.c_void_p), C.c_size_t(16)) == -1


def test_adaptive_rendering_with_the_default_walk(built):
    """Adaptive rendering (active blocks: a subset of the pixels, re-chosen every second pass) over the library's default pipeline --
    dense path state + the 4-wide walk (intersection counters off) -- gives the frame, the block list and the error of the binary walk
    with the counters on."""
    w, h = 256, 192
    scene, camera = scenes.sponza_class(w / h, 8000)
    results = []
    for counters in (False, True):
        vp = ra.Viewport(w, h, seed=14, max_ray_depth=5)
        vp.set_adaptive(True, num_initial_passes=2, min_block_size=8, max_block_size=64, subdivision_treshold=0.05, convergence_treshold=0.01)
        vp.set_renderer(scene, intersection_counters=counters)
        vp.render(camera, 10)
        results.append((vp.sum_buffer(secondary=True), vp.progress(), vp.counters()))
    (a, a2), pa, ca = results[0]
    (b, b2), pb, cb = results[1]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
    assert pa["blocks"] == pb["blocks"] and pa["averageError"] == pb["averageError"] and len(pa["blocks"]) > 1
    assert ca["numRays"] == cb["numRays"] and ca["numRetracedRays"] > 0 and cb["numRetracedRays"] == 0


def test_two_level_scenes_with_the_counters_off(built, monkeypatch):
    """The library's default (intersection counters off) on two-level scenes -- a mesh among analytic shapes and an area light, the Cornell
    box, an instanced mesh beside a sphere: k_trace_wide2 (a 4-wide top-level tree over 4-wide mesh trees, rt_trace_wide2.inl), which hands
    the rays it does not decide to the binary-tree kernel; and with RTGPU_WIDE2=0 the binary walk without the counting code, where a wave's
    idle lanes take over subtrees of its longest any-hit rays at the end of a launch.  Images and ray counters are the oracle's either way."""
    w, h = 192, 108
    for wide2 in ("1", "0", "1 any-hit nearest first"):
        monkeypatch.setenv("RTGPU_WIDE2", wide2[0])
        monkeypatch.setenv("RTGPU_ANYHIT_FAR_FIRST", "0" if "nearest" in wide2 else "1")   # (round 6: any-hit rays walk the farthest entered child first by default)
        wide2 = wide2[0]
        scene, camera = scene_zoo.mesh_scene(w / h, triangles=20000)
        out = run_quant(scene, camera, w, h, passes=3, max_ray_depth=6)
        assert_quant_identical(*out)
        traced = out[2]["numRays"] + out[2]["numShadowRays"]
        assert (0 < out[2]["numRetracedRays"] < 0.02 * traced) if wide2 == "1" else out[2]["numRetracedRays"] == 0, (out[2]["numRetracedRays"], traced)
        assert out[2]["numAnalyticHits"] > 0 and out[2]["numMeshHits"] > 0
        out = run_quant(scene, camera, w, h, passes=2, max_ray_depth=6, light_sampling_all=True)
        assert_quant_identical(*out)
        scene, camera = scenes.cornell_box(w / h)
        assert_quant_identical(*run_quant(scene, camera, w, h, passes=3, max_ray_depth=6))
        assert_quant_identical(*run_quant(scene, camera, w, h, passes=2, max_ray_depth=4, min_russian_roulette_depth=9))
        # two instances of one mesh (they share a tree), a sphere and a two-triangle quad mesh whose tree is a single leaf
        pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(4000, seed=3)
        s = ra.Scene()
        mats = [s.add_material("diffuse", c) for _, c in scenes.SPONZA_MATERIALS]
        metal = s.add_material("roughMetal", (0.9, 0.8, 0.6), roughness=0.2)
        s.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
        s.add_mesh(pos * np.float32(0.2), idx, nrm, tan, uv, mat, mats, transform=ra.transform_from_euler((0.0, 1.0, 0.0), (0.0, 30.0, 0.0)))
        s.add_sphere(0.8, ra.transform_from_euler((-6.0, 1.5, 0.5), (0.0, 0.0, 0.0)), metal)
        quad_pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
        quad_nrm = np.tile(np.array([[0, 1, 0]], np.float32), (4, 1)); quad_tan = np.tile(np.array([[1, 0, 0]], np.float32), (4, 1))
        s.add_mesh(quad_pos, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), quad_nrm, quad_tan, np.zeros((4, 2), np.float32), np.zeros(2, np.uint32), [metal],
                   transform=ra.transform_from_euler((-8.0, 0.6, 0.5), (0.0, 0.0, 0.0)))
        s.add_background_light((1.0, 1.5, 2.0))
        s.add_area_light("rect", [1.0, 1.0], (30.0, 28.0, 25.0), ra.transform_from_euler((-7.0, 6.0, 0.5), (90.0, 0.0, 0.0)))
        s.build()
        cam = ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), w / h, 65.0)
        assert_quant_identical(*run_quant(s, cam, w, h, passes=2, max_ray_depth=6))
        assert_quant_identical(*run_quant(s, cam, w, h, passes=2, max_ray_depth=5, light_sampling_all=True, dimensions=128))


def test_wide_and_exact_traversal_agree_at_full_size(built, monkeypatch):
    """1920x1080, the benchmark's mesh, depth 8: the frame of the 4-wide walk equals the frame of the binary-tree kernel bit for bit, ray
    counters included; the exact re-trace serves well under 1 % of the rays."""
    w, h, depth, passes = 1920, 1080, 8, 2
    scene, camera = scenes.sponza_class(w / h)
    frames = []
    for wide in ("1", "0"):
        monkeypatch.setenv("RTGPU_WIDE", wide)
        vp = ra.Viewport(w, h, seed=515, max_ray_depth=depth)
        vp.set_renderer(scene)
        assert ra.rtgpu_lib().rtgpu_set_intersection_counters(vp.device_context(), 0) == 0
        vp.render(camera, passes)
        frames.append((vp.sum_buffer(secondary=True), vp.counters()))
    (a, a2), ca = frames[0]
    (b, b2), cb = frames[1]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
    for n in NOT_INTERSECTION:
        assert ca[n] == cb[n], n
    traced = ca["numRays"] + ca["numShadowRays"]
    assert 0 < ca["numRetracedRays"] < 0.01 * traced and cb["numRetracedRays"] == 0
    print("retraced %d of %d rays" % (ca["numRetracedRays"], traced))


def test_binary_walk_with_the_counters_off_bit_exact(built, monkeypatch):
    """Intersection counters off (the reference's default) on a single-mesh scene, whose tree the device holds in breadth-first order, with the 4-wide
    walk switched off (RTGPU_WIDE=0): the reference's own binary walk alone gives the oracle's image and ray counters, and hands nothing over."""
    w, h = 128, 72
    monkeypatch.setenv("RTGPU_WIDE", "0")      # the binary-tree walk, not the 4-wide one
    scene, camera = scenes.sponza_class(w / h, 60000)
    a = run_quant(scene, camera, w, h, passes=3, max_ray_depth=8)
    assert_quant_identical(*a)
    assert a[2]["numRetracedRays"] == 0


def test_dense_and_slot_per_pixel_path_state_agree(built, monkeypatch, walk):
    """LightSamplingStrategy::Single runs with dense path state (survivors compacted into a second arena every bounce, finished paths
    parked per pixel, zombies for the last pending shadow ray); RTGPU_NO_DENSE=1 keeps every path in its pixel's slot.  Same images,
    same counters -- on the Cornell box (analytic shapes, area light hits, specular chains), a mesh under two lights (light picking
    from the per-pixel generator), with Russian roulette off and a short depth limit (every path ends as a zombie or at the limit), for
    the plain "Path Tracer", and with more passes than one batch."""
    cases = [(scenes.cornell_box, dict(max_ray_depth=6), 5), (lambda a: scene_zoo.mesh_scene(a, triangles=8000), dict(max_ray_depth=8), 3),
             (lambda a: scenes.sponza_class(a, 20000), dict(max_ray_depth=2, min_russian_roulette_depth=9), 19),
             # LightSamplingStrategy::All with a handful of lights is dense too (every vertex carries one request per light)
             (lambda a: scenes.sponza_class(a, 20000), dict(max_ray_depth=7, light_sampling_all=True, dimensions=128), 3),
             (lambda a: scene_zoo.mesh_scene(a, triangles=8000), dict(max_ray_depth=5, light_sampling_all=True, dimensions=128, min_russian_roulette_depth=9), 2)]
    w, h = 96, 54
    for make, args, passes in cases:
        scene, camera = make(w / h)
        outs = []
        for no_dense in ("0", "1"):
            monkeypatch.setenv("RTGPU_NO_DENSE", no_dense)
            outs.append(run_both(scene, camera, w, h, passes, walk=walk, **args))
        monkeypatch.delenv("RTGPU_NO_DENSE")
        assert_identical(*outs[0]); assert_identical(*outs[1])
        assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
