"""Writes the DDS fixtures (own data): one small file per header variant Bitmap::LoadDDS (Core/Utils/BitmapDDS.cpp) distinguishes.
Payload = a byte pattern derived from the file index.  Regenerate: python tests/golden/dds/make_dds_fixture.py"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
DDPF_FOURCC, DDPF_RGB, DDPF_LUMINANCE = 0x4, 0x40, 0x20000


def fourcc(s):
    return struct.unpack("<I", s.encode())[0]


def header(w, h, flags, four=0, bits=0, masks=(0, 0, 0, 0)):
    pf = struct.pack("<8I", 32, flags, four, bits, *masks)
    return struct.pack("<I", 0x20534444) + struct.pack("<7I", 124, 0x1007, h, w, 0, 0, 0) + b"\0" * 44 + pf + struct.pack("<5I", 0x1000, 0, 0, 0, 0)


def dx10(fmt):
    return struct.pack("<5I", fmt, 3, 0, 1, 0)


# (name, pixel-format flags, fourCC, bit count, masks, DX10 dxgi format or None, bytes per 8x8 image)
VARIANTS = [
    ("bgra8", DDPF_RGB, 0, 32, (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000), None, 256),
    ("rgba8", DDPF_RGB, 0, 32, (0x000000FF, 0x0000FF00, 0x00FF0000, 0xFF000000), None, 256),
    ("r32f_masks", DDPF_RGB, 0, 32, (0xFFFFFFFF, 0, 0, 0), None, 256),
    ("rg16", DDPF_RGB, 0, 32, (0xFFFF, 0xFFFF0000, 0, 0), None, 256),
    ("b5g6r5", DDPF_RGB, 0, 16, (0xF800, 0x07E0, 0x001F, 0), None, 128),
    ("r16f", DDPF_FOURCC, 111, 0, (0, 0, 0, 0), None, 128),
    ("rg16f", DDPF_FOURCC, 112, 0, (0, 0, 0, 0), None, 256),
    ("rgba16f", DDPF_FOURCC, 113, 0, (0, 0, 0, 0), None, 512),
    ("r32f", DDPF_FOURCC, 114, 0, (0, 0, 0, 0), None, 256),
    ("rg32f", DDPF_FOURCC, 115, 0, (0, 0, 0, 0), None, 512),
    ("rgba32f", DDPF_FOURCC, 116, 0, (0, 0, 0, 0), None, 1024),
    ("rgba16", DDPF_FOURCC, 36, 0, (0, 0, 0, 0), None, 512),
    ("dxt1", DDPF_FOURCC, fourcc("DXT1"), 0, (0, 0, 0, 0), None, 32),
    ("ati1", DDPF_FOURCC, fourcc("ATI1"), 0, (0, 0, 0, 0), None, 32),
    ("bc4u", DDPF_FOURCC, fourcc("BC4U"), 0, (0, 0, 0, 0), None, 32),
    ("ati2", DDPF_FOURCC, fourcc("ATI2"), 0, (0, 0, 0, 0), None, 64),
    ("bc5u", DDPF_FOURCC, fourcc("BC5U"), 0, (0, 0, 0, 0), None, 64),
    ("dxt5_unsupported", DDPF_FOURCC, fourcc("DXT5"), 0, (0, 0, 0, 0), None, 64),
    ("l8", DDPF_LUMINANCE, 0, 8, (0xFF, 0, 0, 0), None, 64),
    ("l8a8_8bit", DDPF_LUMINANCE, 0, 8, (0xFF, 0, 0, 0xFF00), None, 128),
    ("l16", DDPF_LUMINANCE, 0, 16, (0xFFFF, 0, 0, 0), None, 128),
    ("l8a8", DDPF_LUMINANCE, 0, 16, (0xFF, 0, 0, 0xFF00), None, 128),
] + [("dx10_%d" % f, DDPF_FOURCC, fourcc("DX10"), 0, (0, 0, 0, 0), f, n) for f, n in
     ((10, 512), (34, 256), (54, 128), (2, 1024), (6, 768), (16, 512), (41, 256), (67, 256), (26, 256), (87, 256), (91, 256), (49, 128), (61, 64),
      (85, 128), (11, 512), (35, 256), (56, 128), (71, 32), (72, 32), (80, 32), (83, 64), (98, 64))]

if __name__ == "__main__":
    for k, (name, flags, four, bits, masks, dxgi, nbytes) in enumerate(VARIANTS):
        payload = bytes(((37 * k + 11 * i + (i >> 3)) & 0xFF) for i in range(nbytes))
        # keep float payloads finite: clear the top exponent bit of every fourth / second byte pattern is not needed for a loader test
        with open(os.path.join(HERE, "%02d_%s.dds" % (k, name)), "wb") as f:
            f.write(header(8, 8, flags, four, bits, masks) + (dx10(dxgi) if dxgi is not None else b"") + payload)
    with open(os.path.join(HERE, "truncated.dds"), "wb") as f:
        f.write(header(8, 8, DDPF_RGB, 0, 32, (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)) + b"\1" * 100)
    print("wrote %d files" % (len(VARIANTS) + 1))
