"""Adaptive rendering against the REFERENCE'S OWN Viewport (tests/golden/adaptive_kat.bin, written by oracle/_ref/ref_render --adaptive-kat:
Viewport.cpp is one of the reference's translation units that build here).  The reference's BuildInitialBlocksList / ComputeBlockError /
UpdateBlocksList (Viewport.cpp:552-581, :618-733) ran eight rounds on synthetic sum buffers; recorded per round: the buffers, every block's
error, the block list after the update (splits, drops, the swap-and-pop order), `converged`, `activePixels`.

* the oracle's rto_block_error must return the reference's errors BIT FOR BIT (the device's rtgpu_compute_block_errors is held against the
  oracle in tests/test_gpu_parity.py);
* the host mirror's rt::Viewport must build the same initial list and, fed the reference's errors, walk to the same list every round."""
import ctypes as C
import os
import struct

import numpy as np

import oracle_lib
import raytracer_amd as ra

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adaptive_kat.bin")


def load():
    raw = open(PATH, "rb").read()
    magic, w, h, rounds, initial, min_block, max_block, _ = struct.unpack_from("<8I", raw, 0)
    assert magic == 0x314B4152
    subdivision, convergence = struct.unpack_from("<2f", raw, 32)
    off = 40

    def blocks():
        nonlocal off
        n = struct.unpack_from("<I", raw, off)[0]; off += 4
        b = np.frombuffer(raw, np.uint32, 4 * n, off).reshape(n, 4); off += 16 * n
        return [tuple(int(v) for v in row) for row in b]

    out = dict(w=w, h=h, settings=dict(num_initial_passes=initial, min_block_size=min_block, max_block_size=max_block, subdivision_treshold=subdivision,
                                       convergence_treshold=convergence), initial=blocks(), rounds=[])
    for _ in range(rounds):
        passes, before = struct.unpack_from("<2I", raw, off); off += 8
        s = np.frombuffer(raw, np.float32, w * h * 3, off).reshape(h, w, 3); off += 12 * w * h
        s2 = np.frombuffer(raw, np.float32, w * h * 3, off).reshape(h, w, 3); off += 12 * w * h
        errors = np.frombuffer(raw, np.float32, before, off).copy(); off += 4 * before
        after = blocks()
        converged, active = struct.unpack_from("<fI", raw, off); off += 8
        out["rounds"].append(dict(passes=passes, sum=s, secondary=s2, errors=errors, after=after, converged=converged, active=active))
    assert off == len(raw)
    return out


def test_block_errors_and_block_lists_match_the_reference_viewport(built):
    k = load()
    w, h = k["w"], k["h"]
    o = oracle_lib.lib()
    o.rto_block_error.restype = C.c_float
    vp = ra.Viewport(w, h, seed=1)
    vp.set_adaptive(True, **k["settings"])     # SetRenderingParams + Reset -> BuildInitialBlocksList
    assert vp.progress()["blocks"] == k["initial"]
    blocks = k["initial"]
    splits = drops = 0
    for r in k["rounds"]:
        s, s2 = np.ascontiguousarray(r["sum"]), np.ascontiguousarray(r["secondary"])
        mine = np.array([o.rto_block_error(s.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.c_uint32(r["passes"]),
                                           C.c_uint32(b[0]), C.c_uint32(b[1]), C.c_uint32(b[2]), C.c_uint32(b[3])) for b in blocks], dtype=np.float32)
        assert np.array_equal(mine.view(np.uint32), r["errors"].view(np.uint32)), (r["passes"], mine, r["errors"])
        n = ra.host_lib().rth_viewport_kat_update_blocks(vp._h, C.c_uint32(r["passes"]), r["errors"].ctypes.data_as(C.c_void_p), C.c_uint32(len(blocks)))
        p = vp.progress()
        assert n == len(r["after"]) and p["blocks"] == r["after"], r["passes"]
        assert np.float32(p["converged"]).view(np.uint32) == np.float32(r["converged"]).view(np.uint32) and p["activePixels"] == r["active"]
        area = lambda bs: sum((b[1] - b[0]) * (b[3] - b[2]) for b in bs)
        if area(r["after"]) < area(blocks):
            drops += 1
        if len(r["after"]) > len(blocks):
            splits += 1
        blocks = r["after"]
    assert splits >= 3 and drops >= 3      # the file really exercises both branches
