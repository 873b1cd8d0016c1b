"""Bloom: Bitmap::GaussianBlur as Viewport::PerformPostProcess drives it.

CPU: the oracle's restatement against tests/golden/bloom_kat.bin, produced by the reference's own Bitmap::GaussianBlur
(oracle/ref_harness/kat_gen.cpp::genBloom) -- bit-exact, including the reference's horizontal pass returning the
second-to-last box blur.  GPU: rtgpu_postprocess with bloomFactor > 0 against the oracle's front buffer -- identical."""
import ctypes as C
import os

import numpy as np
import pytest

import kat_io
import oracle_lib


def bloom_input(count):
    """kat_gen.cpp::bloomInput for i in [0, count)"""
    i = np.arange(count, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15); h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    base = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where((h & np.uint64(0x3F)) == 0, base * np.float32(400.0), base * np.float32(2.0)).astype(np.float32)


def test_gaussian_blur_matches_the_reference(built):
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "bloom_kat.bin"), dtype=np.uint32)
    assert raw[0] == 0x4D4F4C42
    off = 2
    lib = oracle_lib.lib()
    for _ in range(int(raw[1])):
        w, h, levels, step = (int(v) for v in raw[off:off + 4])
        sigma = float(raw[off + 4:off + 5].view(np.float32)[0])
        off += 5
        img = bloom_input(w * h * 3).reshape(h, w, 3).copy()
        for _level in range(levels):
            assert lib.rto_gaussian_blur(img.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(w), C.c_uint32(h), C.c_float(sigma), C.c_uint32(8)) == 0
            sigma = float(np.float32(sigma) * np.float32(2.5))
            want_sum = int(raw[off]) | (int(raw[off + 1]) << 32)
            off += 2
            lattice = img[::step, ::step, :]
            want = raw[off:off + lattice.size].reshape(lattice.shape)
            off += lattice.size
            assert np.array_equal(lattice.view(np.uint32), want), (w, h, _level)
            assert int(img.view(np.uint32).astype(np.uint64).sum()) & 0xFFFFFFFFFFFFFFFF == want_sum
    assert off == raw.size


@pytest.mark.gpu
def test_front_buffer_with_bloom_matches_the_oracle(built):
    import raytracer_amd as ra
    from raytracer_amd import scenes
    w, h = 256, 200
    scene, camera = scenes.cornell_box(w / h)
    vp = ra.Viewport(w, h, seed=3, max_ray_depth=4)
    vp.set_renderer(scene)
    vp.render(camera, 6)
    img = vp.sum_buffer()
    for bloom, tonemapper in ((0.35, 3), (1.0, 0)):
        got = vp.front_buffer(dithering=0.0, bloom=bloom, tonemapper=tonemapper)
        p = ra.RtPostprocessParams()
        for k in range(4):
            p.colorFilter[k] = 1.0
        p.exposure, p.contrast, p.saturation, p.ditheringStrength, p.bloomFactor = 0.0, 0.8, 0.98, 0.0, bloom
        p.tonemapper, p.numPasses, p.ditherSeed = tonemapper, 6, 0
        want = np.zeros((h, w), dtype=np.uint32)
        assert oracle_lib.lib().rto_postprocess_bloom(img.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(w), C.c_uint32(h), C.byref(p),
                                                      want.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        assert np.array_equal(got, want), int(np.count_nonzero(got != want))
        plain = vp.front_buffer(dithering=0.0, bloom=0.0, tonemapper=tonemapper)
        assert np.count_nonzero(plain != got) > 0.5 * w * h   # the light's halo reaches most of the image
    # sizes the reference's blur would read / write out of bounds for are refused
    small = ra.Viewport(64, 48, seed=3)
    small.set_renderer(scene)
    small.render(camera, 1)
    with pytest.raises(RuntimeError):
        small.front_buffer(bloom=0.5)
    assert small.front_buffer(bloom=0.0).shape == (48, 64)
