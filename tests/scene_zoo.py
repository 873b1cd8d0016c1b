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
