"""raytracer_amd -- MI355X-native PathTracerMIS core behind the reference's Scene/Viewport API.

Python is plumbing only: ctypes bindings over
  * lib/librtgpu.so               the C-ABI of include/rtgpu.h (hand-written HIP wavefront path tracer)
  * lib/libraytracer_amd_host.so  the C++ mirror of the reference's host API (rt::Scene, rt::Viewport ...)
                                  through its flat ``rth_*`` facade (host/src/c_api.cpp)

There is NO CPU fallback: creating a renderer without the HIP library / a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.path.join(_PKG_DIR, "lib")
DATA_DIR = os.path.join(_PKG_DIR, "data")
os.environ.setdefault("RT_DATA_DIR", DATA_DIR)

RTGPU_LIB_PATH = os.path.join(_LIB_DIR, "librtgpu.so")
HOST_LIB_PATH = os.path.join(_LIB_DIR, "libraytracer_amd_host.so")


class BuildError(RuntimeError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise BuildError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "raytracer_amd has no CPU fallback." % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_rtgpu = None
_host = None


def rtgpu_lib():
    """The device C-ABI library (include/rtgpu.h)."""
    global _rtgpu
    if _rtgpu is None:
        _rtgpu = _load(RTGPU_LIB_PATH)
        _rtgpu.rtgpu_last_error.restype = C.c_char_p
        _rtgpu.rtgpu_abi_version.restype = C.c_uint32
    return _rtgpu


def host_lib():
    """The C++ host mirror (rt::Scene / rt::Viewport ...) through its rth_* facade."""
    global _host
    if _host is None:
        rtgpu_lib()
        _host = _load(HOST_LIB_PATH)
        h = _host
        for name in ("rth_scene_create", "rth_camera_create", "rth_viewport_create", "rth_viewport_device_ctx"):
            getattr(h, name).restype = C.c_void_p
        h.rth_scene_desc.restype = C.POINTER(RtSceneDesc)
        h.rth_viewport_passes_finished.restype = C.c_uint32
    return _host


# --------------------------------------------------------------------------------------------------
# ctypes mirrors of the PODs in include/rtgpu.h
# --------------------------------------------------------------------------------------------------
class RtNode(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("childIndex", C.c_uint32), ("max", C.c_float * 3), ("leaves", C.c_uint32)]


class RtMesh(C.Structure):
    _fields_ = [("firstNode", C.c_uint32), ("numNodes", C.c_uint32), ("firstTriangle", C.c_uint32), ("numTriangles", C.c_uint32),
                ("firstVertex", C.c_uint32), ("numVertices", C.c_uint32), ("_pad", C.c_uint32 * 2)]


class RtObject(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("invTransform", C.c_float * 16), ("objectKind", C.c_uint32), ("shapeKind", C.c_uint32),
                ("materialIndex", C.c_uint32), ("meshIndex", C.c_uint32), ("lightIndex", C.c_uint32), ("_pad", C.c_uint32 * 3),
                ("shapeParam", C.c_float * 4), ("shapeParam2", C.c_float * 4)]


class RtLight(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("invTransform", C.c_float * 16), ("color", C.c_float * 4), ("type", C.c_uint32),
                ("flags", C.c_uint32), ("shapeKind", C.c_uint32), ("isDelta", C.c_uint32), ("cosAngle", C.c_float), ("texture", C.c_uint32), ("_pad", C.c_float * 2),
                ("shapeParam", C.c_float * 4), ("shapeParam2", C.c_float * 4)]


RT_NO_TEXTURE = 0xFFFFFFFF


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32),
                ("linearSpace", C.c_uint32), ("filter", C.c_uint32), ("numOctaves", C.c_uint32), ("dataOffset", C.c_uint64),
                ("paletteOffset", C.c_uint64), ("colorA", C.c_float * 4), ("colorB", C.c_float * 4), ("mixA", C.c_uint32),
                ("mixB", C.c_uint32), ("mixWeight", C.c_uint32), ("_pad", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("emission", C.c_float * 4), ("baseColor", C.c_float * 4), ("roughness", C.c_float), ("metalness", C.c_float),
                ("IoR", C.c_float), ("K", C.c_float), ("bsdf", C.c_uint32), ("baseColorTexture", C.c_uint32),
                ("emissionTexture", C.c_uint32), ("roughnessTexture", C.c_uint32), ("metalnessTexture", C.c_uint32),
                ("normalMapTexture", C.c_uint32), ("normalMapStrength", C.c_float), ("_pad", C.c_uint32)]


class RtSceneDesc(C.Structure):
    _fields_ = [("abiVersion", C.c_uint32), ("numObjects", C.c_uint32), ("numTopNodes", C.c_uint32), ("numLights", C.c_uint32),
                ("numGlobalLights", C.c_uint32), ("numMaterials", C.c_uint32), ("numMeshes", C.c_uint32), ("numMeshNodes", C.c_uint32),
                ("numTriangles", C.c_uint32), ("numVertices", C.c_uint32), ("numTextures", C.c_uint32), ("_pad", C.c_uint32),
                ("topNodes", C.POINTER(RtNode)), ("objects", C.POINTER(RtObject)), ("lights", C.POINTER(RtLight)),
                ("globalLights", C.POINTER(C.c_uint32)), ("materials", C.POINTER(RtMaterial)), ("meshes", C.POINTER(RtMesh)),
                ("meshNodes", C.POINTER(RtNode)), ("triangles", C.c_void_p), ("vertexIndices", C.c_void_p),
                ("vertexShading", C.c_void_p), ("blueNoise", C.c_void_p), ("textures", C.POINTER(RtTexture)), ("texelData", C.c_void_p),
                ("texelBytes", C.c_uint64)]


class RtPostprocessParams(C.Structure):
    _fields_ = [("colorFilter", C.c_float * 4), ("exposure", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float),
                ("ditheringStrength", C.c_float), ("bloomFactor", C.c_float), ("tonemapper", C.c_uint32), ("numPasses", C.c_uint32),
                ("ditherSeed", C.c_uint32)]


class RtBlock(C.Structure):
    _fields_ = [("minX", C.c_uint32), ("maxX", C.c_uint32), ("minY", C.c_uint32), ("maxY", C.c_uint32)]


class RtCamera(C.Structure):
    _fields_ = [("localToWorld", C.c_float * 16), ("aspectRatio", C.c_float), ("tanHalfFoV", C.c_float), ("dofEnable", C.c_uint32),
                ("bokehShape", C.c_uint32), ("focalPlaneDistance", C.c_float), ("aperture", C.c_float), ("barrelDistortionConstFactor", C.c_float),
                ("barrelDistortionVariableFactor", C.c_float),
                ("worldToScreen", C.c_float * 16)]


class RtMultiInfo(C.Structure):
    _fields_ = [("numDevices", C.c_uint32), ("gatherMode", C.c_uint32), ("gatherReason", C.c_uint32), ("reasonDevice", C.c_int32), ("reasonError", C.c_int32),
                ("devices", C.c_int32 * 16), ("peerAccess", C.c_uint32 * 16), ("reserved", C.c_uint32), ("gathers", C.c_uint64),
                ("lastGatherMs", C.c_double), ("totalGatherMs", C.c_double)]


def multi_info(ctx):
    """rtgpu_get_multi_info as a dict: how a (multi-device) context gathers the peers' tiles at read-back, and why."""
    m = RtMultiInfo()
    if rtgpu_lib().rtgpu_get_multi_info(ctx, C.byref(m)) != 0:
        raise RuntimeError("rtgpu_get_multi_info failed")
    n = int(m.numDevices)
    return {"numDevices": n, "gatherMode": ["none", "peer-kernel", "staged-copy"][m.gatherMode], "gatherReason": ["", "RTGPU_MULTI_STAGED=1", "hipDeviceCanAccessPeer: no", "hipDeviceEnablePeerAccess failed"][m.gatherReason],
            "reasonDevice": int(m.reasonDevice), "reasonError": int(m.reasonError), "devices": [int(m.devices[i]) for i in range(n)], "peerAccess": [bool(m.peerAccess[i]) for i in range(n)],
            "gathers": int(m.gathers), "lastGatherMs": float(m.lastGatherMs), "totalGatherMs": float(m.totalGatherMs)}


class RtPassParams(C.Structure):
    _fields_ = [("camera", RtCamera), ("seed", C.POINTER(C.c_uint32)), ("numDimensions", C.c_uint32), ("useBlueNoise", C.c_uint32),
                ("sampleOffset", C.c_float * 2), ("passIndex", C.c_uint32), ("maxRayDepth", C.c_uint32),
                ("minRussianRouletteDepth", C.c_uint32), ("lightSamplingStrategy", C.c_uint32),
                ("lightSamplingWeight", C.c_float * 4), ("bsdfSamplingWeight", C.c_float * 4), ("rngKey", C.c_uint64 * 2)]


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays", "numRayBoxTests",
                                          "numPassedRayBoxTests", "numRayTriangleTests", "numPassedRayTriangleTests",
                                          "numMeshHits", "numAnalyticHits", "numShadowRayBoxTests",
                                          "numShadowRayTriangleTests", "numRetracedRays")] + [("_reserved", C.c_uint64 * 3)]


COUNTER_NAMES = ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays", "numRayBoxTests", "numPassedRayBoxTests",
                 "numRayTriangleTests", "numPassedRayTriangleTests", "numMeshHits", "numAnalyticHits", "numShadowRayBoxTests",
                 "numShadowRayTriangleTests", "numRetracedRays")

BSDF_NAMES = ("null", "diffuse", "roughDiffuse", "dielectric", "roughDielectric", "metal", "roughMetal", "plastic", "roughPlastic")


def load_blue_noise():
    """128*128*4 uint16 blue-noise table (Data/BlueNoise128_RGBA16.dat of the reference, a data asset)."""
    return np.fromfile(os.path.join(DATA_DIR, "BlueNoise128_RGBA16.dat"), dtype=np.uint16)


def _f(values, n):
    arr = (C.c_float * n)()
    for i, v in enumerate(values):
        arr[i] = float(v)
    return arr


def _color(c):
    c = list(c)
    if len(c) == 3:
        c = c + [0.0]   # Vector4(x, y, z) leaves w = 0, like the reference's constructors / JSON loader
    return _f(c, 4)


def transform_from_euler(translation=(0.0, 0.0, 0.0), orientation_deg=(0.0, 0.0, 0.0)):
    """4x4 row-major local->world matrix from translation + Euler angles in degrees (reference JSON convention)."""
    out = (C.c_float * 16)()
    host_lib().rth_transform_from_euler(_f(translation, 3), _f(orientation_deg, 3), out)
    return out


_IDENTITY = _f([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], 16)


class Scene:
    """rt::Scene: add objects, BuildBVH(), then hand it to a Viewport (reference: Core/Scene/Scene.h)."""

    def __init__(self):
        self._h = C.c_void_p(host_lib().rth_scene_create())
        self._keep = []
        self.built = False
        self.calls = []     # what was added, in order: ("material", {...}), ("sphere", {...}), ... (plain data; lets a tool rebuild the scene)

    def __del__(self):
        try:
            if self._h:
                host_lib().rth_scene_destroy(self._h)
        except Exception:
            pass

    def add_material(self, bsdf="diffuse", base_color=(0.7, 0.7, 0.7), emission=(0.0, 0.0, 0.0), roughness=0.1, metalness=0.0,
                     ior=1.5, k=4.0):
        mid = host_lib().rth_material_create(self._h, bsdf.encode(), _color(base_color), _color(emission), C.c_float(roughness),
                                             C.c_float(metalness), C.c_float(ior), C.c_float(k))
        if mid < 0:
            raise ValueError("unknown BSDF name %r" % bsdf)
        self.calls.append(("material", dict(bsdf=bsdf, base_color=tuple(base_color)[:3], emission=tuple(emission)[:3], roughness=roughness, metalness=metalness, ior=ior, k=k)))
        return mid

    def add_sphere(self, radius, transform=None, material=-1):
        host_lib().rth_add_sphere(self._h, C.c_float(radius), transform or _IDENTITY, int(material))
        self.calls.append(("sphere", dict(radius=radius, transform=list(transform or _IDENTITY), material=int(material))))

    def add_box(self, size, transform=None, material=-1):
        host_lib().rth_add_box(self._h, _f(size, 3), transform or _IDENTITY, int(material))
        self.calls.append(("box", dict(size=tuple(size), transform=list(transform or _IDENTITY), material=int(material))))

    def add_rect(self, size, transform=None, material=-1, tex_scale=(1.0, 1.0)):
        host_lib().rth_add_rect(self._h, _f(size, 2), _f(tex_scale, 2), transform or _IDENTITY, int(material))
        self.calls.append(("rect", dict(size=tuple(size), tex_scale=tuple(tex_scale), transform=list(transform or _IDENTITY), material=int(material))))

    def add_mesh(self, positions, indices, normals=None, tangents=None, tex_coords=None, material_indices=None, materials=(),
                 transform=None, default_material=-1):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)

        def opt(a, w):
            if a is None:
                return None, None
            arr = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, w)
            if arr.shape[0] != pos.shape[0]:
                raise ValueError("per-vertex array has the wrong length")
            return arr, arr.ctypes.data_as(C.POINTER(C.c_float))

        nrm, nrm_p = opt(normals, 3)
        tan, tan_p = opt(tangents, 3)
        uv, uv_p = opt(tex_coords, 2)
        mi, mi_p = None, None
        if material_indices is not None:
            mi = np.ascontiguousarray(material_indices, dtype=np.uint32).reshape(-1)
            mi_p = mi.ctypes.data_as(C.POINTER(C.c_uint32))
        mats = (C.c_int * max(1, len(materials)))(*materials)
        r = host_lib().rth_add_mesh(self._h, C.c_uint32(pos.shape[0]), C.c_uint32(idx.shape[0]), pos.ctypes.data_as(C.POINTER(C.c_float)),
                                    nrm_p, tan_p, uv_p, idx.ctypes.data_as(C.POINTER(C.c_uint32)), mi_p, C.c_uint32(len(materials)), mats,
                                    transform or _IDENTITY, int(default_material))
        if r != 0:
            raise ValueError("mesh rejected (code %d)" % r)
        self.calls.append(("mesh", dict(positions=pos, indices=idx, normals=nrm, tangents=tan, tex_coords=uv, material_indices=mi, materials=tuple(materials),
                                        transform=list(transform or _IDENTITY), material=int(default_material))))

    def add_area_light(self, shape, params, color, transform=None):
        kind = {"sphere": 0, "box": 1, "rect": 2, "plane": 2}[shape]
        p = list(params) + [0.0] * (4 - len(params))
        if host_lib().rth_add_light_area(self._h, kind, _f(p, 4), _color(color), transform or _IDENTITY) != 0:
            raise ValueError("bad area light")
        self.calls.append(("area_light", dict(shape=kind, params=tuple(p), color=tuple(color)[:3], transform=list(transform or _IDENTITY))))

    def add_background_light(self, color, texture=None):
        if texture is None:
            host_lib().rth_add_light_background(self._h, _color(color))
        elif host_lib().rth_add_light_background_textured(self._h, _color(color), int(texture)) != 0:
            raise ValueError("bad environment map texture")
        self.calls.append(("background_light", dict(color=tuple(color)[:3], texture=texture)))

    # ---- ingestion (helpers::LoadScene / LoadMesh of the reference's Demo, in the C++ host mirror) ------------------
    def load_json(self, path, data_path="", camera=None):
        """helpers::LoadScene: adds the objects / lights of a JSON scene file (and sets `camera`, a Camera, if given).
        data_path is prepended to the mesh / texture paths of the file (Options::dataPath)."""
        if host_lib().rth_load_scene(self._h, camera._h if camera is not None else None, str(path).encode(), str(data_path).encode()) != 0:
            raise ValueError("LoadScene failed: %s" % path)
        return self

    # ---- textures (ITexture of the reference; evaluated on the device) --------------------------------------------
    FORMATS = dict(R8_UNorm=1, R8G8_UNorm=2, B8G8R8_UNorm=3, B8G8R8A8_UNorm=4, R8G8B8A8_UNorm=5, B8G8R8A8_UNorm_Palette=6, B5G6R5_UNorm=7,
                   R16_UNorm=8, R16G16_UNorm=9, R16G16B16A16_UNorm=10, R32_Float=11, R32G32_Float=12, R32G32B32_Float=13,
                   R32G32B32A32_Float=14, R11G11B10_Float=15, R16_Half=16, R16G16_Half=17, R16G16B16_Half=18, R16G16B16A16_Half=19,
                   R9G9B9E5_SharedExp=20, BC1=21, BC4=22, BC5=23)
    FILTERS = dict(nearest=0, bilinear=1, smoothstep=2)

    def add_bitmap_texture(self, pixels, fmt, linear_space=True, filter="smoothstep", palette=None, size=None):
        """pixels: C-contiguous numpy array of shape (height, width[, channels]) whose dtype/channels match `fmt`
        (uint8, uint16, float16 or float32); rows are tightly packed."""
        a = np.ascontiguousarray(pixels)
        if size is not None:            # block-compressed data: `pixels` is the raw block stream, size = (width, height) in texels
            w, h = size
            stride = 0
        else:
            h, w = a.shape[0], a.shape[1]
            stride = a.strides[0]
        tid = host_lib().rth_texture_bitmap(self._h, C.c_uint32(w), C.c_uint32(h), C.c_uint32(self.FORMATS[fmt]), a.ctypes.data_as(C.c_void_p),
                                            C.c_uint32(stride), 1 if linear_space else 0, self.FILTERS[filter])
        if tid < 0:
            raise ValueError("bad bitmap texture")
        self.calls.append(("bitmap_texture", dict(id=int(tid), width=int(w), height=int(h), format=fmt, stride=int(stride), linear_space=bool(linear_space),
                                                  filter=filter, pixels=a, has_palette=palette is not None)))
        if palette is not None:
            pal = np.ascontiguousarray(palette, dtype=np.uint8).reshape(-1, 4)
            if host_lib().rth_texture_set_palette(self._h, tid, pal.ctypes.data_as(C.c_void_p), C.c_uint32(pal.shape[0])) != 0:
                raise ValueError("bad palette")
        return tid

    def add_noise_texture(self, color_a, color_b, octaves=1):
        return host_lib().rth_texture_noise(self._h, _color(color_a), _color(color_b), C.c_uint32(octaves))

    def add_mix_texture(self, texture_a, texture_b, weight):
        tid = host_lib().rth_texture_mix(self._h, int(texture_a), int(texture_b), int(weight))
        if tid < 0:
            raise ValueError("bad mix texture children")
        return tid

    def add_checkerboard_texture(self, color_a, color_b):
        return host_lib().rth_texture_checkerboard(self._h, _color(color_a), _color(color_b))

    def add_const_texture(self, color):
        return host_lib().rth_texture_const(self._h, _color(color))

    def set_material_texture(self, material, slot, texture, strength=1.0):
        """slot: 'baseColor' | 'emission' | 'roughness' | 'metalness' | 'normal' (strength = normalMapStrength)"""
        slots = dict(baseColor=0, emission=1, roughness=2, metalness=3, normal=4)
        if host_lib().rth_material_set_texture(self._h, int(material), slots[slot], int(texture), C.c_float(strength)) != 0:
            raise ValueError("bad material / texture id")
        self.calls.append(("material_texture", dict(material=int(material), slot=slot, texture=int(texture), strength=float(strength))))

    def add_directional_light(self, color, angle_rad=0.2, transform=None):
        host_lib().rth_add_light_directional(self._h, _color(color), C.c_float(angle_rad), transform or _IDENTITY)
        self.calls.append(("directional_light", dict(color=tuple(color)[:3], angle=float(angle_rad), transform=list(transform or _IDENTITY))))

    def add_point_light(self, color, transform=None):
        host_lib().rth_add_light_point(self._h, _color(color), transform or _IDENTITY)
        self.calls.append(("point_light", dict(color=tuple(color)[:3], transform=list(transform or _IDENTITY))))

    def add_spot_light(self, color, angle_rad, transform=None):
        host_lib().rth_add_light_spot(self._h, _color(color), C.c_float(angle_rad), transform or _IDENTITY)
        self.calls.append(("spot_light", dict(color=tuple(color)[:3], angle=float(angle_rad), transform=list(transform or _IDENTITY))))

    def build(self):
        if host_lib().rth_scene_build(self._h) != 0:
            raise RuntimeError("Scene::BuildBVH failed")
        self.built = True
        return self

    @property
    def desc(self):
        """Pointer to the flat RtSceneDesc (valid until the scene is rebuilt / destroyed)."""
        return host_lib().rth_scene_desc(self._h)


class Camera:
    """rt::Camera (reference: Core/Scene/Camera.h)."""

    def __init__(self, translation=(0.0, 0.0, 0.0), orientation_deg=(0.0, 0.0, 0.0), aspect=1.0, fov_deg=20.0):
        self._h = C.c_void_p(host_lib().rth_camera_create())
        self.settings = dict(dof=False, focal_plane_distance=2.0, aperture=0.1)
        self.set_transform(translation, orientation_deg)
        self.set_perspective(aspect, np.float32(fov_deg) / np.float32(180.0) * np.float32(3.14159265359))

    def __del__(self):
        try:
            if self._h:
                host_lib().rth_camera_destroy(self._h)
        except Exception:
            pass

    def set_transform(self, translation, orientation_deg=(0.0, 0.0, 0.0)):
        host_lib().rth_camera_set_transform(self._h, _f(translation, 3), _f(orientation_deg, 3))
        self.settings.update(translation=tuple(translation), orientation_deg=tuple(orientation_deg))

    def set_perspective(self, aspect, fov_rad):
        host_lib().rth_camera_set_perspective(self._h, C.c_float(aspect), C.c_float(fov_rad))
        self.settings.update(aspect=float(aspect), fov_rad=float(fov_rad))

    def set_dof(self, enable, focal_plane_distance=2.0, aperture=0.1):
        host_lib().rth_camera_set_dof(self._h, int(bool(enable)), C.c_float(focal_plane_distance), C.c_float(aperture))
        self.settings.update(dof=bool(enable), focal_plane_distance=float(focal_plane_distance), aperture=float(aperture))

    def set_lens(self, bokeh_shape=0, barrel_const=0.01, barrel_variable=0.0):
        """DOFSettings::bokehShape (0 circle, 1 hexagon, 2 square) and the barrel-distortion factors of rt::Camera."""
        host_lib().rth_camera_set_lens(self._h, C.c_uint32(bokeh_shape), C.c_float(barrel_const), C.c_float(barrel_variable))
        self.settings.update(bokeh_shape=int(bokeh_shape), barrel_const=float(barrel_const), barrel_variable=float(barrel_variable))


class Viewport:
    """rt::Viewport driving the GPU "Path Tracer MIS" renderer (reference: Core/Rendering/Viewport.h)."""

    def __init__(self, width, height, seed=None, dimensions=64, use_blue_noise=True, anti_aliasing_spread=0.5, max_ray_depth=20,
                 min_russian_roulette_depth=1, light_sampling_all=False):
        self._h = C.c_void_p(host_lib().rth_viewport_create())
        self.width, self.height = int(width), int(height)
        self._scene = None
        self.has_renderer = False
        if host_lib().rth_viewport_set_params(self._h, C.c_uint32(dimensions), int(bool(use_blue_noise)), C.c_float(anti_aliasing_spread),
                                              C.c_uint32(max_ray_depth), C.c_uint32(min_russian_roulette_depth), int(bool(light_sampling_all))) != 0:
            raise ValueError("invalid rendering params")
        if seed is not None:
            host_lib().rth_viewport_set_seed(self._h, C.c_uint64(seed))
        if host_lib().rth_viewport_resize(self._h, C.c_uint32(width), C.c_uint32(height)) != 0:
            raise ValueError("invalid viewport size")

    def __del__(self):
        try:
            if self._h:
                host_lib().rth_viewport_destroy(self._h)
        except Exception:
            pass

    def set_renderer(self, scene, name="Path Tracer MIS", device=-1, devices=None, intersection_counters=False):
        """CreateRenderer(name, scene) + SetRenderer.  Raises when the GPU renderer cannot be created.  `devices`: a list of HIP device indices
        for ONE renderer over several GPUs of the node (SetRendererDevices -> rtgpu_create_multi; an index may repeat).
        `intersection_counters`: False = the library's (and the reference's, Core/Config.h:4) default; True turns the box / triangle test
        counters on, which routes every ray through the reference's binary walk (parity tests compare those counters too)."""
        self._scene = scene
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            host_lib().rth_set_renderer_devices(arr, C.c_uint32(len(devices)))
        try:
            r = host_lib().rth_viewport_set_renderer(self._h, scene._h, name.encode(), int(device))
        finally:
            if devices is not None:
                host_lib().rth_set_renderer_devices(None, C.c_uint32(0))
        if r != 0:
            err = rtgpu_lib().rtgpu_last_error()
            raise RuntimeError("CreateRenderer(%r) failed (%d): %s" % (name, r, err.decode() if err else ""))
        self.has_renderer = True
        if intersection_counters:
            rtgpu_lib().rtgpu_set_intersection_counters(self.device_context(), 1)
        self.reset()

    def set_vcm(self, max_path_length=10, use_vertex_connection=True, use_vertex_merging=True, initial_merging_radius=0.02,
                min_merging_radius=0.02, merging_radius_multiplier=1.0, bsdf_weight=1.0, light_weight=1.0, vertex_connecting_weight=1.0,
                camera_connecting_weight=1.0, vertex_merging_weight=1.0):
        """The public members of rt::VertexConnectionAndMerging (renderer name "VCM"); takes effect with the next pass."""
        w = (C.c_float * 5)(bsdf_weight, light_weight, vertex_connecting_weight, camera_connecting_weight, vertex_merging_weight)
        r = host_lib().rth_viewport_set_vcm(self._h, C.c_uint32(max_path_length), int(bool(use_vertex_connection)), int(bool(use_vertex_merging)),
                                            C.c_float(initial_merging_radius), C.c_float(min_merging_radius), C.c_float(merging_radius_multiplier), w)
        if r != 0:
            raise RuntimeError("set_vcm: the viewport's renderer is not \"VCM\"")

    def set_debug_mode(self, mode):
        """DebugRenderer::mRenderingMode (renderer name "Debug"): 0 CameraLight, 1 TriangleID, 2 Depth, 3 Position, 4 Normals, 5 Tangents,
        6 Bitangents, 7 TexCoords, 8 BaseColor, 9 Emission, 10 Roughness, 11 Metalness, 12 IoR."""
        if host_lib().rth_viewport_set_debug_mode(self._h, C.c_uint32(mode)) != 0:
            raise RuntimeError("set_debug_mode: the viewport's renderer is not \"Debug\" or the mode is unknown")

    def vcm_num_photons(self):
        n = C.c_uint32(0)
        ctx = C.c_void_p(host_lib().rth_viewport_device_ctx(self._h))
        if rtgpu_lib().rtgpu_vcm_num_photons(ctx, C.byref(n)) != 0:
            raise RuntimeError(rtgpu_lib().rtgpu_last_error().decode())
        return int(n.value)

    def set_shard(self, rank, world_size):
        if host_lib().rth_viewport_set_shard(self._h, C.c_uint32(rank), C.c_uint32(world_size)) != 0:
            raise RuntimeError("set_shard failed")

    def reset(self):
        host_lib().rth_viewport_reset(self._h)

    def render(self, camera, passes=1):
        if host_lib().rth_viewport_render(self._h, camera._h, C.c_uint32(passes)) != 0:
            raise RuntimeError("Viewport::Render failed: %s" % (rtgpu_lib().rtgpu_last_error() or b"").decode())

    def next_pass_params(self, camera):
        """Per-pass constants (Halton seeds, AA offset ...) exactly as Render() would use them; advances the state."""
        p = RtPassParams()
        if host_lib().rth_viewport_next_pass_params(self._h, camera._h, C.byref(p)) != 0:
            raise RuntimeError("NextPassParams failed")
        # copy the seeds: the pointer refers to storage reused by the next call
        seeds = np.ctypeslib.as_array(p.seed, shape=(p.numDimensions,)).copy()
        p._seed_keepalive = seeds
        p.seed = seeds.ctypes.data_as(C.POINTER(C.c_uint32))
        return p

    def render_pass_with(self, params):
        """Submit one pass with explicit constants (used by the parity tests)."""
        if host_lib().rth_viewport_render_pass_with(self._h, C.byref(params)) != 0:
            raise RuntimeError("render pass failed: %s" % (rtgpu_lib().rtgpu_last_error() or b"").decode())

    def set_adaptive(self, enable=True, num_initial_passes=10, min_block_size=4, max_block_size=256, subdivision_treshold=0.005,
                     convergence_treshold=0.0001):
        """RenderingParams::adaptiveSettings (Core/Rendering/Context.h:35-43); resets the viewport."""
        if host_lib().rth_viewport_set_adaptive(self._h, int(bool(enable)), C.c_uint32(num_initial_passes), C.c_uint32(min_block_size),
                                                C.c_uint32(max_block_size), C.c_float(subdivision_treshold), C.c_float(convergence_treshold)) != 0:
            raise ValueError("bad adaptive settings")

    def progress(self):
        """RenderingProgress + the active block list [(minX, maxX, minY, maxY), ...]."""
        err, conv, pixels = C.c_float(), C.c_float(), C.c_uint32()
        n = host_lib().rth_viewport_progress(self._h, C.byref(err), C.byref(conv), C.byref(pixels), None, 0)
        blocks = np.zeros((max(n, 1), 4), dtype=np.uint32)
        host_lib().rth_viewport_progress(self._h, None, None, None, blocks.ctypes.data_as(C.c_void_p), C.c_uint32(n))
        return dict(averageError=err.value, converged=conv.value, activePixels=pixels.value, blocks=[tuple(int(v) for v in b) for b in blocks[:n]])

    def front_buffer(self, exposure=0.0, contrast=0.8, saturation=0.98, dithering=0.005, tonemapper=3, color_filter=(1.0, 1.0, 1.0, 1.0),
                     dither_seed=0, bloom=0.0):
        """Viewport::PostProcessTile on the device (defaults = PostprocessParams(), Core/Rendering/PostProcess.cpp:6-14):
        the (H, W) uint32 0x00RRGGBB front buffer of the passes rendered so far."""
        p = RtPostprocessParams()
        for k in range(4):
            p.colorFilter[k] = color_filter[k]
        p.exposure, p.contrast, p.saturation, p.ditheringStrength, p.bloomFactor = exposure, contrast, saturation, dithering, bloom
        p.tonemapper, p.numPasses, p.ditherSeed = int(tonemapper), max(1, self.passes_finished), int(dither_seed)
        out = np.zeros((self.height, self.width), dtype=np.uint32)
        if rtgpu_lib().rtgpu_postprocess(self.device_context(), C.byref(p), out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("postprocess failed: %s" % (rtgpu_lib().rtgpu_last_error() or b"").decode())
        return out

    def device_context(self):
        return C.c_void_p(host_lib().rth_viewport_device_ctx(self._h))

    def sum_buffer(self, secondary=False):
        """Accumulated float3 image, shape (H, W, 3); row y as in the reference's sum bitmap.  Synchronises."""
        s = np.zeros((self.height, self.width, 3), dtype=np.float32)
        s2 = np.zeros((self.height, self.width, 3), dtype=np.float32) if secondary else None
        host_lib().rth_viewport_read_sum(self._h, s.ctypes.data_as(C.POINTER(C.c_float)),
                                         s2.ctypes.data_as(C.POINTER(C.c_float)) if secondary else None)
        return (s, s2) if secondary else s

    def counters(self):
        out = (C.c_uint64 * 16)()
        host_lib().rth_viewport_counters(self._h, out)
        d = {n: int(out[i]) for i, n in enumerate(COUNTER_NAMES)}
        if self.has_renderer:   # a statistic of the device library, not part of the reference's RayTracingCounters
            raw = RtCounters()
            if rtgpu_lib().rtgpu_get_counters(self.device_context(), C.byref(raw)) == 0:
                d["numRetracedRays"] = int(raw.numRetracedRays)
                d["numUntrustedRays"], d["numStackOverflowRays"], d["diag2"] = int(raw._reserved[0]), int(raw._reserved[1]), int(raw._reserved[2])
        return d

    @property
    def passes_finished(self):
        return int(host_lib().rth_viewport_passes_finished(self._h))
