#!/usr/bin/env python3
"""Writes the OBJ / MTL / BMP ingestion fixture (tests/golden/obj/fixture.obj, fixture.mtl, albedo.bmp).  Own data,
deterministic.  It exercises what decides the numbers helpers::LoadMesh produces: number syntaxes of tinyobjloader's
tryParseDouble (plain, signed, exponents, long fractions), v/vt/vn index forms incl. negative (relative) indices,
triangles, convex and concave polygons (ear clipping), faces without normals / without texture coordinates, a
degenerate triangle (dropped), shared vertices (de-duplication), several materials incl. an unknown usemtl name,
groups / objects, a diffuse texture (BMP) and Windows line endings in the MTL."""
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(20260928)


def fmt(x, style):
    if style == 0:
        return "%.6f" % x
    if style == 1:
        return "%.9e" % x
    if style == 2:
        return "%+.4f" % x
    if style == 3:
        return "%.12f" % x
    return repr(float(np.float32(x)))


lines = ["# ingestion fixture", "mtllib fixture.mtl", "o first"]
verts, uvs, norms = [], [], []


def v(p):
    verts.append(p); lines.append("v " + " ".join(fmt(c, (len(verts) + i) % 5) for i, c in enumerate(p)))
    return len(verts)


def vt(p):
    uvs.append(p); lines.append("vt " + " ".join(fmt(c, (len(uvs) + i) % 4) for i, c in enumerate(p)))
    return len(uvs)


def vn(p):
    norms.append(p); lines.append("vn " + " ".join(fmt(c, (len(norms) + i) % 3) for i, c in enumerate(p)))
    return len(norms)


# a bumpy 12 x 9 grid with normals and uvs, quads and triangles, three materials
NX, NY = 12, 9
grid = {}
for j in range(NY + 1):
    for i in range(NX + 1):
        x, y = i * 0.37 - 2.0, j * 0.41 - 1.5
        z = 0.3 * math.sin(1.3 * x) * math.cos(0.9 * y) + 0.02 * rng.standard_normal()
        n = np.array([-0.39 * math.cos(1.3 * x) * math.cos(0.9 * y), 0.27 * math.sin(1.3 * x) * math.sin(0.9 * y), 1.0])
        n = n / np.linalg.norm(n) * (1.0 + 0.3 * rng.uniform())     # not unit length: LoadMesh normalises
        grid[(i, j)] = (v((x, y, z)), vt((i / NX * 2.0, j / NY * 3.0 - 0.5)), vn(tuple(n)))
lines.append("g bumpy")
mats = ["red_textured", "green", "blue emissive"]
for j in range(NY):
    for i in range(NX):
        a, b, c, d = grid[(i, j)], grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]
        if (i + j) % 5 == 0:
            lines.append("usemtl " + mats[(i * 7 + j) % 3])
        f = lambda p: "%d/%d/%d" % p
        if (i * 3 + j) % 4 == 0:
            lines.append("f %s %s %s" % (f(a), f(b), f(c))); lines.append("f %s %s %s" % (f(a), f(c), f(d)))
        else:
            lines.append("f %s %s %s %s" % (f(a), f(b), f(c), f(d)))

# polygons without normals (face normal fallback), relative indices, concave shapes
lines += ["o polygons", "usemtl green"]
for k in range(6):
    nverts = 5 + k
    base = []
    for t in range(nverts):
        ang = 2.0 * math.pi * t / nverts
        r = 0.8 if (t % 2 == 0 or k % 2 == 0) else 0.3        # odd k: star shaped (concave)
        base.append(v((3.0 + k * 1.9 + r * math.cos(ang), r * math.sin(ang), 0.1 * k + 0.05 * math.sin(3 * ang))))
        vt((0.5 + 0.5 * math.cos(ang), 0.5 + 0.5 * math.sin(ang)))
    if k % 3 == 0:
        lines.append("f " + " ".join("%d/%d" % (i - len(verts) - 1, i - len(verts) - 1) for i in base))   # negative = relative (v and vt advance together here)
    elif k % 3 == 1:
        lines.append("f " + " ".join("%d" % i for i in base))              # no uvs, no normals
    else:
        lines.append("usemtl does_not_exist")
        lines.append("f " + " ".join("%d/%d" % (i, len(uvs) - nverts + 1 + t) for t, i in enumerate(base)))
        lines.append("usemtl red_textured")

# faces with normals but no uvs (v//vn), a degenerate triangle, a tiny triangle
lines += ["g mixed", "s 1"]
a = v((0.0, 4.0, 0.0)); b = v((1.0, 4.0, 0.2)); c = v((0.5, 5.0, 0.1)); n0 = vn((0.0, 0.1, 1.0))
lines.append("f %d//%d %d//%d %d//%d" % (a, n0, b, n0, c, n0))
lines.append("f %d//%d %d//%d %d//%d" % (a, n0, a, n0, b, n0))              # degenerate: dropped
d = v((0.0, 4.0, 0.0002)); e = v((0.0003, 4.0, 0.0)); lines.append("f %d %d %d" % (a, d, e))   # shorter than MinEdgeLength: dropped
lines += ["s off", "f %d %d %d" % (a, c, b), ""]
open(os.path.join(HERE, "fixture.obj"), "w").write("\n".join(lines))

mtl = ["# materials", "newmtl red_textured", "Ka 0 0 0", "Kd 0.8 0.15 0.1", "Ks 0.5 0.5 0.5", "Ns 20", "map_Kd -clamp on albedo.bmp", "",
       "newmtl green   ", "Kd 1.5e-1 .8e0 0.2", "Ke 0 0 0", "",
       "newmtl blue emissive", "Kd 0.1 0.2 0.9", "Ke 2.5 3 4.25", "d 1.0", "illum 2"]
open(os.path.join(HERE, "fixture.mtl"), "wb").write(("\r\n".join(mtl) + "\r\n").encode())

# 24-bit BMP, 5 x 3 (rows padded to 16 bytes)
w, h = 5, 3
row = (w * 3 + 3) & ~3
pixels = bytearray()
for y in range(h):
    r = bytearray(rng.randint(0, 256, size=w * 3).astype(np.uint8).tobytes())
    pixels += r + bytes(row - w * 3)
header = struct.pack("<HIHHI", 0x4D42, 54 + len(pixels), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(pixels), 2835, 2835, 0, 0)
open(os.path.join(HERE, "albedo.bmp"), "wb").write(header + bytes(pixels))
print("wrote fixture.obj (%d lines), fixture.mtl, albedo.bmp" % len(lines))
