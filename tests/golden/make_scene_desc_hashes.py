"""Pins the mirror's JSON scene loader (raytracer_amd/host/Demo/SceneLoader.cpp, helpers::LoadScene) to the REFERENCE'S OWN scene files.

Run in the build container (where /root/reference exists):   python tests/golden/make_scene_desc_hashes.py
Loads every /root/reference/Data/TestScenes/*.json through helpers::LoadScene (data path = /root/reference/Data/, as the reference's Demo passes
Options::dataPath) and writes tests/golden/scene_desc_hashes.json: per file whether it loads and, if it does, the flattened RtSceneDesc -- counts,
SHA-256 of the object / light / material / top-level-node arrays (textures excluded), the lights and the camera IN CLEAR.  A file that does not load
records the reason; every such reason is the reference's own failure mode for that file (Demo/SceneLoader.cpp:692-820 returns false on the first
entry that does not parse; a missing asset fails ParseTexture / LoadMesh) or a documented scope limit (CSG).
The fixture is derived data (numbers and hashes), not the text of the reference's files.  tests/test_scene_files.py holds the Python scene builders
(raytracer_amd/scenes.py: CORNELL_BOX, ROUGH_GLASS_SLAB's camera recipe, the Sponza lights) against these entries on every box, and re-derives the
whole file where /root/reference exists."""
import ctypes as C
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REFERENCE_SCENES = "/root/reference/Data/TestScenes"
REFERENCE_DATA = "/root/reference/Data/"
OUT = os.path.join(ROOT, "tests", "golden", "scene_desc_hashes.json")

# why a reference scene file does not load (checked by hand against Demo/SceneLoader.cpp; the loader's log line is matched, not the file's text)
EXPECTED_FAILURES = {
    "dispersion_test.json": "area light without 'shape': the reference's ParseLight returns false too (Demo/SceneLoader.cpp:525-531); the file predates that loader",
    "glossy_refraction_test.json": "area light without 'shape' (Demo/SceneLoader.cpp:525-531)",
    "sds_test.json": "area light without 'shape' (Demo/SceneLoader.cpp:525-531)",
    "small_light_test.json": "area light without 'shape' (Demo/SceneLoader.cpp:525-531)",
    "sphere_light_test.json": "light type 'sphere': the reference builds the AreaLight and drops it (Demo/SceneLoader.cpp:587-597 never assigns `light`), "
                              "then wraps a null light in a LightSceneObject; the mirror refuses the type instead of dereferencing null",
    "shapes_test.json": "a CSG object: outside SURVEY 8's scope (the device path has no CsgShape), refused at load",
    "glass_bunny.json": "missing asset MODELS/bunny.obj (not in the reference checkout): helpers::LoadMesh fails in the reference too",
    "sponza.json": "missing asset MODELS/crytek-sponza/sponza.obj (not in the reference checkout, SURVEY 0.5): helpers::LoadMesh fails in the reference too",
    "texture_test.json": "missing texture assets (TEXTURES/... not in the reference checkout): ParseTexture fails in the reference too",
}


def struct_dict(s):
    out = {}
    for name, ty in s._fields_:
        if name.startswith("_"):
            continue
        v = getattr(s, name)
        out[name] = [float(x) if isinstance(x, float) else int(x) for x in v] if hasattr(v, "__len__") else (float(v) if isinstance(v, float) else int(v))
    return out


def array_sha(ptr, count, ty):
    if not count:
        return hashlib.sha256(b"").hexdigest()
    return hashlib.sha256(C.string_at(C.cast(ptr, C.c_void_p), C.sizeof(ty) * count)).hexdigest()


def describe(scene, camera, ra):
    d = scene.desc.contents
    cam = ra.RtCamera()
    assert ra.host_lib().rth_camera_desc(camera._h, C.byref(cam)) == 0
    return {
        "counts": {k: int(getattr(d, k)) for k in ("numObjects", "numTopNodes", "numLights", "numGlobalLights", "numMaterials", "numMeshes", "numTextures")},
        "sha256": {"objects": array_sha(d.objects, d.numObjects, ra.RtObject), "lights": array_sha(d.lights, d.numLights, ra.RtLight),
                   "materials": array_sha(d.materials, d.numMaterials, ra.RtMaterial), "topNodes": array_sha(d.topNodes, d.numTopNodes, ra.RtNode)},
        "lights": [struct_dict(d.lights[i]) for i in range(d.numLights)],
        "materials": [struct_dict(d.materials[i]) for i in range(d.numMaterials)],
        "camera": struct_dict(cam),
    }


def load_reference_scene(ra, path, aspect=1.0):
    camera = ra.Camera((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), aspect, 60.0)
    scene = ra.Scene().load_json(path, data_path=REFERENCE_DATA, camera=camera)
    scene.build()
    return scene, camera


def build_fixture(ra):
    out = {}
    for path in sorted(glob.glob(os.path.join(REFERENCE_SCENES, "*.json"))):
        name = os.path.basename(path)
        try:
            scene, camera = load_reference_scene(ra, path)
        except ValueError:
            out[name] = {"loads": False, "why": EXPECTED_FAILURES.get(name, "UNEXPECTED: not among the failures checked against Demo/SceneLoader.cpp")}
            if "missing asset MODELS" in out[name]["why"]:
                # the rest of such a file (materials, lights, camera) is still the reference's description of the scene: loaded with the mesh objects
                # taken out (a temporary copy; only numbers derived from it are kept)
                import tempfile
                with open(path) as f:
                    doc = json.load(f)
                doc["objects"] = [o for o in doc.get("objects", []) if o.get("type") != "mesh"]
                with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as tmp:
                    tmp.write(json.dumps(doc))
                try:
                    scene, camera = load_reference_scene(ra, tmp.name)
                    out[name]["without_mesh_objects"] = describe(scene, camera, ra)
                finally:
                    os.unlink(tmp.name)
            continue
        entry = describe(scene, camera, ra)
        entry["loads"] = True
        out[name] = entry
    return out


if __name__ == "__main__":
    import raytracer_amd as ra
    fixture = build_fixture(ra)
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1, sort_keys=True)
        f.write("\n")
    loaded = [n for n, e in fixture.items() if e["loads"]]
    print("%d scene files, %d load; not loading: %s" % (len(fixture), len(loaded), ", ".join(n for n in fixture if n not in loaded)))
    unexpected = [n for n, e in fixture.items() if not e["loads"] and e["why"].startswith("UNEXPECTED")]
    if unexpected:
        raise SystemExit("unexpected load failures: %s" % unexpected)
