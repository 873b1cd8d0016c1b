"""Scenes shared by the CPU (oracle-only) and GPU parity tests."""
import numpy as np

import raytracer_amd as ra
from raytracer_amd import scenes


def all_lights_scene(aspect):
    """Every light type + every BSDF on analytic shapes (LightSamplingStrategy::All friendly)."""
    s = ra.Scene()
    mats = {name: s.add_material(name, (0.8, 0.6, 0.4) if i % 2 else (0.3, 0.7, 0.9), roughness=0.05 + 0.1 * i, ior=1.3 + 0.05 * i)
            for i, name in enumerate(ra.BSDF_NAMES) if name != "null"}
    emissive = s.add_material("null", (0.0, 0.0, 0.0), emission=(0.5, 0.25, 0.125))
    names = list(mats)
    k = 0
    for iz in range(3):
        for ix in range(3):
            t = ra.transform_from_euler((-3.0 + 3.0 * ix, -1.0, -3.0 + 3.0 * iz), (10.0 * ix, 25.0 * iz, 5.0))
            if k % 2 == 0:
                s.add_sphere(0.9, t, mats[names[k % len(names)]])
            else:
                s.add_box((0.7, 0.9, 0.6), t, mats[names[k % len(names)]])
            k += 1
    s.add_rect((8.0, 8.0), ra.transform_from_euler((0.0, -2.0, 0.0), (-90.0, 0.0, 0.0)), mats["diffuse"])
    s.add_sphere(0.4, ra.transform_from_euler((0.0, 1.5, 0.0)), emissive)
    s.add_area_light("rect", [1.0, 0.5], (6.0, 6.0, 6.0), ra.transform_from_euler((0.0, 4.0, 0.0), (90.0, 0.0, 0.0)))
    s.add_area_light("sphere", [0.5], (4.0, 2.0, 1.0), ra.transform_from_euler((3.5, 2.0, 1.0)))
    s.add_area_light("box", [0.3, 0.2, 0.4], (1.0, 3.0, 5.0), ra.transform_from_euler((-3.5, 2.5, -1.0), (20.0, 30.0, 0.0)))
    s.add_point_light((20.0, 18.0, 15.0), ra.transform_from_euler((2.0, 3.0, -3.0)))
    s.add_spot_light((40.0, 40.0, 60.0), 0.6, ra.transform_from_euler((-2.0, 3.5, 3.0), (0.0, 0.0, 0.0)))
    s.add_directional_light((2.0, 2.0, 1.5), 0.1, ra.transform_from_euler((0.0, 0.0, 0.0), (60.0, 30.0, 0.0)))
    s.add_directional_light((3.0, 2.5, 2.0), 0.001, ra.transform_from_euler((0.0, 0.0, 0.0), (70.0, -40.0, 0.0)))
    s.add_background_light((0.2, 0.3, 0.5))
    s.build()
    cam = ra.Camera((0.5, 2.5, 9.0), (12.0, 180.0, 0.0), aspect, 55.0)
    return s, cam


def mesh_scene(aspect, triangles=20000, with_analytic=True):
    """Sponza-class mesh (small) + a few analytic instances so that both BVH levels are exercised."""
    pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(triangles, seed=5)
    s = ra.Scene()
    mats = [s.add_material("diffuse", c) for _, c in scenes.SPONZA_MATERIALS]
    s.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
    if with_analytic:
        glass = s.add_material("dielectric", (1.0, 1.0, 1.0))
        metal = s.add_material("roughMetal", (0.9, 0.8, 0.6), roughness=0.3)
        s.add_sphere(1.2, ra.transform_from_euler((-4.0, 1.2, 0.5)), glass)
        s.add_box((0.8, 1.5, 0.8), ra.transform_from_euler((3.0, 1.5, -1.0), (0.0, 30.0, 0.0)), metal)
        s.add_area_light("rect", [1.5, 1.5], (30.0, 28.0, 25.0), ra.transform_from_euler((0.0, 11.0, 0.0), (90.0, 0.0, 0.0)))
    s.add_background_light((1.0, 1.5, 2.0))
    s.add_directional_light((20000.0, 19000.0, 18000.0), np.float32(1.0) / np.float32(180.0) * np.float32(3.14159265359),
                            ra.transform_from_euler((0.0, 0.0, 0.0), (80.0, 20.0, 0.0)))
    s.build()
    cam = ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), aspect, 65.0)
    return s, cam


def textured_scene(aspect, triangles=6000, seed=11, skip=()):
    """Textures on every slot of the shading path: a mesh whose 8 materials carry bitmap base-colour / roughness /
    metalness / emission textures in different texel formats, colour spaces and filters, normal maps (bitmap and
    checkerboard), analytic shapes with checkerboard / const textures, and an HDR environment map on the background
    light (BackgroundLight::mTexture)."""
    rng = np.random.RandomState(seed)
    pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(triangles, seed=5)
    s = ra.Scene()

    def rgba8(h, w):
        return rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8)

    t_albedo = s.add_bitmap_texture(rgba8(32, 48), "B8G8R8A8_UNorm", linear_space=False, filter="smoothstep")
    t_albedo2 = s.add_bitmap_texture(rng.randint(0, 256, size=(7, 5, 3)).astype(np.uint8), "B8G8R8_UNorm", linear_space=False, filter="bilinear")
    t_rough = s.add_bitmap_texture(rng.randint(0, 256, size=(16, 16)).astype(np.uint8), "R8_UNorm", filter="nearest")
    t_metal = s.add_bitmap_texture(rng.randint(0, 65536, size=(9, 13)).astype(np.uint16), "R16_UNorm", filter="bilinear")
    t_emit = s.add_bitmap_texture(rng.uniform(0.0, 1.0, size=(8, 8, 3)).astype(np.float32), "R32G32B32_Float", filter="smoothstep")
    nm = rng.uniform(0.35, 0.65, size=(24, 24, 4)).astype(np.float32)
    t_normal = s.add_bitmap_texture(nm, "R32G32B32A32_Float", filter="smoothstep")
    t_normal_h = s.add_bitmap_texture(rng.uniform(0.3, 0.7, size=(12, 10, 2)).astype(np.float16), "R16G16_Half", filter="bilinear")
    t_check = s.add_checkerboard_texture((0.9, 0.2, 0.1), (0.1, 0.3, 0.9, 1.0))
    t_const = s.add_const_texture((0.5, 0.75, 1.0, 0.0))
    t_noise = s.add_noise_texture((0.9, 0.8, 0.7), (0.2, 0.1, 0.4), octaves=4)
    t_mix = s.add_mix_texture(t_albedo, t_check, t_noise)
    bc1 = rng.randint(0, 256, size=(16 // 4) * (8 // 4) * 8).astype(np.uint8)
    t_bc1 = s.add_bitmap_texture(bc1, "BC1", linear_space=False, filter="bilinear", size=(16, 8))
    env = rng.uniform(0.0, 3.0, size=(16, 32, 4)).astype(np.float16)
    t_env = s.add_bitmap_texture(env, "R16G16B16A16_Half", filter="smoothstep")

    kinds = ["diffuse", "roughDiffuse", "roughPlastic", "roughMetal", "diffuse", "plastic", "roughDielectric", "diffuse"]
    mats = []
    for i, (_, c) in enumerate(scenes.SPONZA_MATERIALS):
        m = s.add_material(kinds[i], c, roughness=0.4, metalness=0.3 if i == 3 else 0.0,
                           emission=(0.6, 0.5, 0.4) if i == 5 else (0.0, 0.0, 0.0))
        mats.append(m)
    if 0 not in skip: s.set_material_texture(mats[0], "baseColor", t_albedo)
    if 1 not in skip: s.set_material_texture(mats[0], "normal", t_normal, 0.8)
    if 2 not in skip: s.set_material_texture(mats[1], "baseColor", t_albedo2)
    if 3 not in skip: s.set_material_texture(mats[1], "roughness", t_rough)
    if 4 not in skip: s.set_material_texture(mats[2], "roughness", t_rough)
    if 5 not in skip: s.set_material_texture(mats[2], "baseColor", t_check)
    if 6 not in skip: s.set_material_texture(mats[3], "metalness", t_metal)
    if 7 not in skip: s.set_material_texture(mats[3], "normal", t_normal_h, 1.0)
    if 8 not in skip: s.set_material_texture(mats[4], "normal", t_check, 0.5)
    if 9 not in skip: s.set_material_texture(mats[5], "emission", t_emit)
    if 10 not in skip: s.set_material_texture(mats[6], "roughness", t_metal)
    if 11 not in skip: s.set_material_texture(mats[7], "baseColor", t_const)
    if 15 not in skip: s.set_material_texture(mats[4], "baseColor", t_mix)
    if 16 not in skip: s.set_material_texture(mats[6], "baseColor", t_bc1)
    if 17 not in skip: s.set_material_texture(mats[5], "roughness", t_noise)
    s.add_mesh(pos, idx, nrm, tan, uv, mat, mats)

    ball = s.add_material("roughPlastic", (0.8, 0.8, 0.8), roughness=0.3)
    if 12 not in skip: s.set_material_texture(ball, "baseColor", t_check)
    if 13 not in skip: s.set_material_texture(ball, "normal", t_normal, 1.0)
    s.add_sphere(1.2, ra.transform_from_euler((-4.0, 1.2, 0.5)), ball)
    crate = s.add_material("diffuse", (1.0, 1.0, 1.0))
    if 14 not in skip: s.set_material_texture(crate, "baseColor", t_albedo)
    s.add_box((0.8, 1.5, 0.8), ra.transform_from_euler((3.0, 1.5, -1.0), (0.0, 30.0, 0.0)), crate)
    s.add_area_light("rect", [1.5, 1.5], (30.0, 28.0, 25.0), ra.transform_from_euler((0.0, 11.0, 0.0), (90.0, 0.0, 0.0)))
    s.add_background_light((1.0, 1.2, 1.5), texture=None if 'env' in skip else t_env)
    s.build()
    cam = ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), aspect, 65.0)
    return s, cam


def many_lights_scene(aspect, num_point_lights=14):
    """More than 8 lights (under LightSamplingStrategy::All the NEE requests past the eighth take the per-lane append
    path of k_shade), some of them black (zero radiance: no shadow ray), a one-triangle mesh and a mesh-free object."""
    s = ra.Scene()
    grey = s.add_material("roughDiffuse", (0.7, 0.7, 0.7), roughness=0.4)
    s.add_rect((6.0, 6.0), ra.transform_from_euler((0.0, -1.0, 0.0), (-90.0, 0.0, 0.0)), grey)
    s.add_sphere(0.8, ra.transform_from_euler((0.0, -0.2, 0.0)), s.add_material("plastic", (0.9, 0.3, 0.2)))
    pos = np.array([[-2.0, -0.9, 1.0], [2.0, -0.9, 1.0], [0.0, 1.5, -1.5]], dtype=np.float32)
    s.add_mesh(pos, np.array([[0, 1, 2]], dtype=np.uint32), None, None, None, None, [], default_material=grey)
    for i in range(num_point_lights):
        a = 2.0 * np.pi * i / num_point_lights
        color = (0.0, 0.0, 0.0) if i % 5 == 4 else (3.0 + i, 4.0, 12.0 - 0.5 * i)
        s.add_point_light(color, ra.transform_from_euler((3.0 * np.cos(a), 2.0 + 0.1 * i, 3.0 * np.sin(a))))
    s.add_area_light("sphere", [0.3], (5.0, 5.0, 4.0), ra.transform_from_euler((0.0, 3.0, 0.0)))
    s.add_background_light((0.05, 0.06, 0.08))
    s.build()
    return s, ra.Camera((0.0, 1.5, 7.0), (8.0, 180.0, 0.0), aspect, 50.0)
