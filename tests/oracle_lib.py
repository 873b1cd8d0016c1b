"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE: the product package
raytracer_amd never imports this."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "_build", "liboracle.so")

_lib = None

KAT = dict(SIN_LANE=1, SINCOS=2, FASTLOG=3, FASTACOS=4, FASTATAN2=5, FLOAT_NORMAL2=6, HEMISPHERE_COS=7, SPHERE=8, CIRCLE=9,
           ORTHO_BASIS=10, FRESNEL_DIELECTRIC=11, FRESNEL_METAL=12, REFRACT3=13, REFLECT3=14, BOX_RAY=20, BOX_RAY_TWOSIDED=21,
           TRIANGLE_RAY=22, MAKE_RAY=23, TRANSFORM_RAY=24, FAST_INVERSE=25, TRANSFORM_SCALED=26, FRAME_COMPOSE=27, SHAPE_INTERSECT=30, SHAPE_SAMPLE=31, SHAPE_PDF=32,
           SHAPE_EVAL=33, LIGHT_ILLUMINATE=40, LIGHT_RADIANCE=41, BSDF_SAMPLE=50, BSDF_EVALUATE=51, CAMERA_RAY=60)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_PATH):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(ORACLE_PATH)
        _lib.rto_sizeof.restype = C.c_uint32
    return _lib


def render_pass(scene_desc_ptr, params, width, height, sum_buf, secondary=None, counters=None, shard=(0, 1), threads=1, plain=False):
    """One pass of the oracle into sum_buf (H, W, 3 float32, accumulated in place).  plain: the renderer "Path Tracer"."""
    if counters is None:
        counters = np.zeros(16, dtype=np.uint64)
    r = (lib().rto_render_pass_plain if plain else lib().rto_render_pass)(scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), C.c_uint32(shard[0]), C.c_uint32(shard[1]),
                              sum_buf.ctypes.data_as(C.POINTER(C.c_float)),
                              secondary.ctypes.data_as(C.POINTER(C.c_float)) if secondary is not None else None,
                              counters.ctypes.data_as(C.POINTER(C.c_uint64)), int(threads))
    if r != 0:
        raise RuntimeError("rto_render_pass failed")
    return counters


def render_pixel(scene_desc_ptr, params, width, height, x, y):
    out = (C.c_float * 4)()
    lib().rto_render_pixel(scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), C.c_uint32(x), C.c_uint32(y), out, None)
    return np.array(out[:], dtype=np.float32)


def render_pixel_paths(scene_desc_ptr, params, width, height, x, y, capacity=64):
    """The vertices of ONE pixel's path as the reference's PathDebugData hook records them: (numVertices, 28) float32 (see RenderCtx)."""
    out = (C.c_float * 4)()
    vertices = np.zeros((capacity, 28), dtype=np.float32)
    n = C.c_uint32(0)
    lib().rto_render_pixel_paths(scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), C.c_uint32(x), C.c_uint32(y), out, None,
                                 vertices.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(capacity), C.byref(n))
    assert n.value <= capacity
    return vertices[:n.value]


def kat(func, inputs, out_stride):
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    n, in_stride = inputs.shape
    out = np.zeros((n, out_stride), dtype=np.float32)
    r = lib().rto_kat(int(KAT[func] if isinstance(func, str) else func), inputs.ctypes.data_as(C.POINTER(C.c_float)), int(in_stride),
                      out.ctypes.data_as(C.POINTER(C.c_float)), int(out_stride), int(n))
    if r != 0:
        raise RuntimeError("unknown KAT function %r" % (func,))
    return out


def sampler_ints(seed, blue_noise, use_blue_noise, x, y, count):
    seed = np.ascontiguousarray(seed, dtype=np.uint32)
    ints = np.zeros(count, dtype=np.uint32)
    floats = np.zeros(count, dtype=np.float32)
    bn = blue_noise.ctypes.data_as(C.POINTER(C.c_uint16)) if blue_noise is not None else None
    lib().rto_kat_sampler(seed.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(seed)), bn, C.c_uint32(1 if use_blue_noise else 0),
                          C.c_uint32(x), C.c_uint32(y), C.c_uint32(count), ints.ctypes.data_as(C.POINTER(C.c_uint32)),
                          floats.ctypes.data_as(C.POINTER(C.c_float)))
    return ints, floats


def xoroshiro(s0, s1, count):
    out = np.zeros(count, dtype=np.uint64)
    lib().rto_kat_xoroshiro(C.c_uint64(s0), C.c_uint64(s1), C.c_uint32(count), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


class Vcm:
    """The oracle's VertexConnectionAndMerging restatement (oracle/rto_vcm.h): holds the photons between passes."""

    def __init__(self, max_path_length=10, use_vertex_connection=True, use_vertex_merging=True, initial_merging_radius=0.02,
                 min_merging_radius=0.02, merging_radius_multiplier=1.0, bsdf_weight=1.0, light_weight=1.0, vertex_connecting_weight=1.0,
                 camera_connecting_weight=1.0, vertex_merging_weight=1.0):
        words = np.zeros(28, dtype=np.uint32)
        words[0] = max_path_length; words[1] = int(use_vertex_connection); words[2] = int(use_vertex_merging)
        words[3:6] = np.array([initial_merging_radius, min_merging_radius, merging_radius_multiplier], dtype=np.float32).view(np.uint32)
        for k, wgt in enumerate((bsdf_weight, light_weight, vertex_connecting_weight, camera_connecting_weight, vertex_merging_weight)):
            words[8 + 4 * k:12 + 4 * k] = np.full(4, wgt, dtype=np.float32).view(np.uint32)
        self.settings = words
        lib().rto_vcm_create.restype = C.c_void_p
        lib().rto_vcm_num_photons.restype = C.c_uint32
        self._h = C.c_void_p(lib().rto_vcm_create(words.ctypes.data_as(C.POINTER(C.c_uint32))))
        self.passes = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rto_vcm_destroy(self._h)
            self._h = None

    def num_photons(self):
        return int(lib().rto_vcm_num_photons(self._h))

    def render_pass(self, scene_desc_ptr, params, width, height, sum_buf, secondary=None, light_sum=None, counters=None, shard=(0, 1)):
        """shard = (rank, world): only the pixels of the 64x64 tiles with tile % world == rank (exact for the camera paths with merging off)"""
        if counters is None:
            counters = np.zeros(16, dtype=np.uint64)
        fp = C.POINTER(C.c_float)
        r = lib().rto_vcm_render_pass_tiles(self._h, scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), C.c_uint32(self.passes),
                                            sum_buf.ctypes.data_as(fp), secondary.ctypes.data_as(fp) if secondary is not None else None,
                                            light_sum.ctypes.data_as(fp) if light_sum is not None else None, counters.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            C.c_uint32(shard[0]), C.c_uint32(shard[1]))
        if r != 0:
            raise RuntimeError("rto_vcm_render_pass failed")
        self.passes += 1
        return counters


def render_pass_debug(scene_desc_ptr, params, width, height, mode, sum_buf, secondary=None, counters=None, threads=1):
    """One pass of the oracle's DebugRenderer restatement (renderer "Debug", DebugRenderingMode `mode`)."""
    if counters is None:
        counters = np.zeros(16, dtype=np.uint64)
    fp = C.POINTER(C.c_float)
    r = lib().rto_render_pass_debug(scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), C.c_uint32(mode), sum_buf.ctypes.data_as(fp),
                                    secondary.ctypes.data_as(fp) if secondary is not None else None, counters.ctypes.data_as(C.POINTER(C.c_uint64)), int(threads))
    if r != 0:
        raise RuntimeError("rto_render_pass_debug failed")
    return counters


def light_tracer_pass(scene_desc_ptr, params, width, height, sum_buf, secondary=None, counters=None):
    """One pass of the oracle's LightTracer restatement (renderer "Light Tracer"); single-threaded (the film splats are ordered)."""
    if counters is None:
        counters = np.zeros(16, dtype=np.uint64)
    fp = C.POINTER(C.c_float)
    r = lib().rto_light_tracer_render_pass(scene_desc_ptr, C.byref(params), C.c_uint32(width), C.c_uint32(height), sum_buf.ctypes.data_as(fp),
                                           secondary.ctypes.data_as(fp) if secondary is not None else None, counters.ctypes.data_as(C.POINTER(C.c_uint64)))
    if r != 0:
        raise RuntimeError("rto_light_tracer_render_pass failed")
    return counters


def hash_grid_query(points, radius, queries, capacity=1 << 22):
    """HashGrid::Build + Process (Utils/HashGrid.h) over points[n,3] for queries[m,3]: list of index arrays in visiting order."""
    L = lib()
    points = np.ascontiguousarray(points, dtype=np.float32); queries = np.ascontiguousarray(queries, dtype=np.float32)
    offsets = np.zeros(len(queries) + 1, dtype=np.uint64); indices = np.zeros(capacity, dtype=np.uint32)
    L.rto_hash_grid_query.restype = C.c_uint64
    L.rto_hash_grid_query.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
    total = L.rto_hash_grid_query(points.ctypes.data, len(points), radius, queries.ctypes.data, len(queries), offsets.ctypes.data, indices.ctypes.data, capacity)
    assert total <= capacity
    return [indices[int(offsets[q]):int(offsets[q + 1])] for q in range(len(queries))]


def set_x86_approximations(enable):
    """rto_math.h's x86 approximation mode (FastDivide through _mm_rcp_ss, FastNormalize3 through _mm_rsqrt_ps, as the reference evaluates them).
    Returns (supported, (rcp_ss(3.0) bits, rsqrt_ps(0.7) bits)) -- the signature of this CPU's approximation tables."""
    sig = (C.c_uint32 * 2)()
    ok = lib().rto_set_x86_approximations(int(bool(enable)), sig)
    return bool(ok), (int(sig[0]), int(sig[1]))
